// stand-in for okvis_common/include/okvis/IdProvider.hpp (process-wide id source)
#pragma once
#include <atomic>
#include <cstdint>
namespace okvis {
class IdProvider {
 public:
  static IdProvider& instance() { static IdProvider p; return p; }
  uint64_t newId() { return ++id_; }
 private:
  std::atomic<uint64_t> id_{0};
};
}  // namespace okvis
