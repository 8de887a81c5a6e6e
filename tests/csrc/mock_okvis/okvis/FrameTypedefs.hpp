// stand-in for okvis_common/include/okvis/FrameTypedefs.hpp:60-135 (KeypointIdentifier, MapPoint, containers)
#pragma once
#include <cstdint>
#include <map>
#include <vector>
#include "mock_eigen.hpp"
namespace okvis {
struct KeypointIdentifier {
  explicit KeypointIdentifier(uint64_t fi = 0, size_t ci = 0, size_t ki = 0) : frameId(fi), cameraIndex(ci), keypointIndex(ki) {}
  uint64_t frameId; size_t cameraIndex; size_t keypointIndex;
  bool operator<(const KeypointIdentifier& r) const {
    if (frameId != r.frameId) return frameId < r.frameId;
    if (cameraIndex != r.cameraIndex) return cameraIndex < r.cameraIndex;
    return keypointIndex < r.keypointIndex;
  }
};
struct MapPoint {
  MapPoint() : id(0), quality(0.0), distance(0.0) {}
  uint64_t id; Eigen::Vector4d point; double quality; double distance;
  std::map<okvis::KeypointIdentifier, uint64_t> observations;
};
typedef std::vector<MapPoint> MapPointVector;
typedef std::map<uint64_t, MapPoint> PointMap;
}  // namespace okvis
