// stand-in for okvis_time/include/okvis/Time.hpp:128 (sec / nsec pair)
#pragma once
#include <cstdint>
namespace okvis {
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() = default;
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
};
}  // namespace okvis
