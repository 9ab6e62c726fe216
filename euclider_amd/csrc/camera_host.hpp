/*
 * camera_host.hpp -- Camera::update on the host (rotation maths of the three reference cameras);
 * translation goes through a trace_path callback, which the C ABI binds to the GPU kernel.
 */
#ifndef EUCLIDER_AMD_CAMERA_HOST_HPP
#define EUCLIDER_AMD_CAMERA_HOST_HPP

#include <functional>

#include "../../include/euclider_amd.h"

namespace euclider {

/* Universe::trace_path_unknown: returns 1 = Some, 0 = None, negative = EU_ERR_*.  A null function means
 * "no universe available": the update fails with EU_ERR_NO_DEVICE if it has to move the camera. */
using TracePathFn = std::function<int(const double *location, const double *direction, double distance, double *out_location, double *out_direction)>;

int camera_update(eu_camera *cam, const eu_input *in, const TracePathFn &trace_path);

}  // namespace euclider
#endif
