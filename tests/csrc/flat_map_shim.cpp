// C entry point over svin_amd/csrc/flat_map.hpp for tests/test_flat_map_host.py: the map behind the window's per-observation
// bookkeeping (residual id -> landmark node, landmark id -> handle) against std::unordered_map under a random operation stream.
#include "../../svin_amd/csrc/flat_map.hpp"

#include <unordered_map>

extern "C" {
// ops[i] = {kind (0 set, 1 erase, 2 find), key, value}; returns the index of the first disagreement with std::unordered_map,
// -1 if none; *finalSize = size() at the end
long fm_replay(const uint64_t* ops, long n, uint64_t* finalSize) {
  svin::FlatMap64 m;
  std::unordered_map<uint64_t, uint64_t> ref;
  for (long i = 0; i < n; ++i) {
    const uint64_t kind = ops[3 * i], key = ops[3 * i + 1], val = ops[3 * i + 2];
    if (kind == 0) { m.set(key, val); ref[key] = val; }
    else if (kind == 1) { if (m.erase(key) != (ref.erase(key) == 1)) return i; }
    else {
      uint64_t got = 0;
      const auto it = ref.find(key);
      if (m.find(key, &got) != (it != ref.end())) return i;
      if (it != ref.end() && got != it->second) return i;
      if (m.count(key) != (it != ref.end())) return i;
    }
    if (m.size() != ref.size()) return i;
  }
  for (const auto& kv : ref) {   // every surviving entry is still reachable (the backward-shift deletion kept the probe chains whole)
    uint64_t got = 0;
    if (!m.find(kv.first, &got) || got != kv.second) return n;
  }
  *finalSize = m.size();
  return -1;
}
}
