/*
 * camera_host.cpp -- the reference's Camera::update for PitchYawCamera3, FreeCamera3 and FreeCamera4
 * (d3/entity/camera.rs:94-145,191-245,299-346,396-451; d4/entity/camera.rs:68-136,182-241;
 * util.rs:301-322).  Host code: a handful of flops per frame.  The only part that needs the scene,
 * Universe::trace_path_unknown, is a callback (the C ABI runs it on the GPU).
 *
 * Third-party arithmetic restated here ("parity unpinned"): nalgebra 0.8.2
 * UnitQuaternion::new(axis*angle) / rotate, cross, Matrix4 storage order and products,
 * ApproxEq::approx_eq_ulps (8 ulps); det 0.1.0 `det_copy!` (Laplace expansion along the first row).
 */
#include "camera_host.hpp"

#include <cmath>
#include <cstring>

#include "eu_math.h"

namespace euclider {
namespace {

constexpr double PI_C = 3.14159265358979323846264338327950288;

double dot_n(int D, const double *a, const double *b) { double s = a[0] * b[0]; for (int i = 1; i < D; i++) s = s + a[i] * b[i]; return s; }
double norm_n(int D, const double *a) { return std::sqrt(dot_n(D, a, a)); }
void normalize_n(int D, double *a) { const double n = norm_n(D, a); for (int i = 0; i < D; i++) a[i] = a[i] / n; }
double angle_between_n(int D, const double *a, const double *b) {      /* util.rs:712-722 */
    const double r = eu_acos(dot_n(D, a, b) / (norm_n(D, a) * norm_n(D, b)));
    return (r != r) ? 0.0 : r;
}
void cross3(const double *a, const double *b, double *o) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

struct Quat { double w, x, y, z; };
Quat quat_new(const double *axisangle) {      /* nalgebra UnitQuaternion::new */
    const double sq = dot_n(3, axisangle, axisangle);
    if (sq == 0.0) return Quat{1.0, 0.0, 0.0, 0.0};
    const double ang = std::sqrt(sq);
    const double s = eu_sin(ang / 2.0), c = eu_cos(ang / 2.0);
    const double s_ang = s / ang;
    return Quat{c, axisangle[0] * s_ang, axisangle[1] * s_ang, axisangle[2] * s_ang};
}
void quat_rotate(const Quat &q, double *v) {  /* UnitQuaternion * Vector3: t = 2 (qv x v); v' = t w + qv x t + v */
    const double qv[3] = {q.x, q.y, q.z};
    double t[3], u[3];
    cross3(qv, v, t);
    t[0] = t[0] * 2.0; t[1] = t[1] * 2.0; t[2] = t[2] * 2.0;
    cross3(qv, t, u);
    for (int i = 0; i < 3; i++) v[i] = (t[i] * q.w + u[i]) + v[i];
}
void rotate_about(const double *axis, double angle, double *v) {
    const double aa[3] = {axis[0] * angle, axis[1] * angle, axis[2] * angle};
    quat_rotate(quat_new(aa), v);
}

bool approx_eq_ulps(double a, double b, uint32_t ulps) {   /* nalgebra 0.8.2 ApproxEq<f64>::approx_eq_ulps */
    if (a == b) return true;
    if (std::signbit(a) != std::signbit(b) || a != a || b != b) return false;
    int64_t ia, ib;
    std::memcpy(&ia, &a, 8); std::memcpy(&ib, &b, 8);
    const int64_t d = ia - ib;
    return (d < 0 ? -d : d) < (int64_t)ulps;
}

/* ---- 3-D ------------------------------------------------------------------------------------------ */
void yaw_pitchyaw(double *forward, double *up, double angle) {      /* camera.rs:110-114 */
    const double z[3] = {0.0, 0.0, 1.0};
    rotate_about(z, angle, forward); normalize_n(3, forward);
    rotate_about(z, angle, up); normalize_n(3, up);
}
void pitch_static(double *forward, double *up, double angle, bool snap) {   /* camera.rs:116-137 */
    double axis_h[3];
    cross3(forward, up, axis_h); normalize_n(3, axis_h);
    if (snap) {
        const double z[3] = {0.0, 0.0, 1.0};
        const double result_angle = angle_between_n(3, forward, z);
        if (result_angle < angle) {
            forward[0] = 0.0; forward[1] = 0.0; forward[2] = 1.0;
            cross3(axis_h, forward, up); normalize_n(3, up);
            return;
        } else if (PI_C - result_angle < -angle) {
            forward[0] = -0.0; forward[1] = -0.0; forward[2] = -1.0;
            cross3(axis_h, forward, up); normalize_n(3, up);
            return;
        }
    }
    rotate_about(axis_h, angle, forward); normalize_n(3, forward);
    cross3(axis_h, forward, up); normalize_n(3, up);
}
void left3(const eu_camera *c, double *o) { cross3(c->up, c->forward, o); normalize_n(3, o); }   /* camera.rs:60-62 */

int move_camera(eu_camera *cam, int D, double *direction, double distance, const TracePathFn &trace_path, bool allow_turn) {
    if (dot_n(D, direction, direction) == 0.0) return EU_OK;        /* camera.rs:218, d4:214 */
    const double length = norm_n(D, direction);
    distance *= length;
    normalize_n(D, direction);
    if (!trace_path) return EU_ERR_NO_DEVICE;
    double nl[4] = {0, 0, 0, 0}, nd[4] = {0, 0, 0, 0};
    const int rc = trace_path(cam->location, direction, distance, nl, nd);
    if (rc < 0) return rc;
    if (rc == 0) return EU_OK;                                       /* None: the camera stays */
    const double rotation_scale = angle_between_n(D, direction, nd);
    if (allow_turn) {
        if (!approx_eq_ulps(rotation_scale, 0.0, 8u)) {              /* camera.rs:230-239 */
            double axis[3];
            cross3(direction, nd, axis);
            const double aa[3] = {axis[0] * rotation_scale, axis[1] * rotation_scale, axis[2] * rotation_scale};
            const Quat q = quat_new(aa);
            quat_rotate(q, cam->forward);
            quat_rotate(q, cam->up);
        }
    } else if (!approx_eq_ulps(rotation_scale, 0.0, 4u * 8u)) {
        return EU_ERR_UNIMPLEMENTED;                                 /* d4/entity/camera.rs:227-235 `unimplemented!()` */
    }
    for (int i = 0; i < D; i++) cam->location[i] = nl[i];
    return EU_OK;
}

int update3(eu_camera *cam, const eu_input *in, double sens, double speed, const TracePathFn &trace_path) {
    const bool free_cam = cam->kind == EU_CAMERA_FREE_3;
    const double delta_millis = (double)in->delta_time_ms / 1000.0;
    const double mx = (double)in->delta_mouse_x, my = (double)in->delta_mouse_y;
    if (!free_cam) {                                                 /* PitchYawCamera3::update_rotation, camera.rs:94-108 */
        if (!(mx * mx + my * my <= 0.0)) {
            const double dx = mx * sens, dy = my * sens;
            yaw_pitchyaw(cam->forward, cam->up, -dx);
            pitch_static(cam->forward, cam->up, -dy, true);
        }
    } else {                                                         /* FreeCamera3::update_rotation, camera.rs:299-324 */
        const double dx = mx * sens, dy = my * sens;
        double roll = 0.0;
        if (in->keys & EU_KEY_Q) roll -= 1.0;
        if (in->keys & EU_KEY_E) roll += 1.0;
        roll *= delta_millis * 2.0;
        if (dx != 0.0) { rotate_about(cam->up, -dx, cam->forward); normalize_n(3, cam->forward); }          /* :326-329 */
        if (dy != 0.0) pitch_static(cam->forward, cam->up, -dy, false);
        if (roll != 0.0) { rotate_about(cam->forward, roll, cam->up); normalize_n(3, cam->up); }            /* :331-334 */
    }
    const double distance = speed * delta_millis;
    if (distance == 0.0) return EU_OK;
    double direction[3] = {0.0, 0.0, 0.0}, left[3];
    const double z[3] = {0.0, 0.0, 1.0};
    const double *vertical = free_cam ? cam->up : z;                 /* camera.rs:214-217 vs :419-422 */
    auto add = [&](const double *v, double k) { for (int i = 0; i < 3; i++) direction[i] = k > 0 ? direction[i] + v[i] : direction[i] - v[i]; };
    if (in->keys & EU_KEY_W) add(cam->forward, 1);
    if (in->keys & EU_KEY_S) add(cam->forward, -1);
    if (in->keys & EU_KEY_A) { left3(cam, left); add(left, 1); }
    if (in->keys & EU_KEY_D) { left3(cam, left); add(left, -1); }
    if (in->keys & EU_KEY_LSHIFT) add(vertical, 1);
    if (in->keys & EU_KEY_LCONTROL) add(vertical, -1);
    return move_camera(cam, 3, direction, distance, trace_path, true);
}

/* ---- 4-D ------------------------------------------------------------------------------------------ */
double det3(const double m[3][3]) {           /* first-row Laplace expansion */
    return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
void find_orthonormal_4(const double *a, const double *b, const double *c, double *o) {   /* util.rs:301-308 */
    const double *rows[3] = {a, b, c};
    double res[4];
    for (int k = 0; k < 4; k++) {
        double m[3][3];
        for (int r = 0; r < 3; r++) { int cc = 0; for (int col = 0; col < 4; col++) if (col != k) m[r][cc++] = rows[r][col]; }
        const double minor = det3(m);
        res[k] = (k & 1) ? -minor : minor;
    }
    for (int k = 0; k < 4; k++) o[k] = res[k];
}
void reorthonormalize_4(double *a, double *b, double *c, const double *d) {                 /* util.rs:310-322 */
    find_orthonormal_4(c, b, d, a); normalize_n(4, a);
    find_orthonormal_4(a, c, d, b); normalize_n(4, b);
    find_orthonormal_4(a, d, b, c); normalize_n(4, c);
}
void mat4_vec(const double m[4][4], const double *v, double *o) {      /* row-major m[r][c]; x -> w accumulation */
    double r[4];
    for (int i = 0; i < 4; i++) r[i] = ((m[i][0] * v[0] + m[i][1] * v[1]) + m[i][2] * v[2]) + m[i][3] * v[3];
    for (int i = 0; i < 4; i++) o[i] = r[i];
}

int update4(eu_camera *cam, const eu_input *in, double speed, const TracePathFn &trace_path) {
    const double delta_millis = (double)in->delta_time_ms / 1000.0;
    double angle = 0.0;                                              /* d4/entity/camera.rs:68-126 */
    if (in->keys & EU_KEY_C) angle += 1.0;
    if (in->keys & EU_KEY_M) angle -= 1.0;
    if (angle != 0.0) {
        angle *= delta_millis * 2.0;
        const bool axis[4] = {(in->keys & EU_KEY_I) != 0, (in->keys & EU_KEY_O) != 0, (in->keys & EU_KEY_K) != 0, (in->keys & EU_KEY_L) != 0};
        int count = 0;
        for (int i = 0; i < 4; i++) count += axis[i] ? 1 : 0;
        if (count == 2) {
            /* iter_mut() walks nalgebra's column-major storage, so the reference's `row = index / 4` is the
             * storage COLUMN and `column = index % 4` the storage ROW */
            double rot[4][4];
            const double ca = eu_cos(angle), sa = eu_sin(angle);
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) rot[r][c] = (r == c) ? 1.0 : 0.0;
            for (int index = 0; index < 16; index++) {
                const int rr = index / 4, cc = index % 4;            /* the reference's names */
                if (axis[rr] && axis[cc]) rot[cc][rr] = (rr == cc) ? ca : (rr < cc ? -sa : sa);
            }
            double ana[4];
            find_orthonormal_4(cam->forward, cam->left, cam->up, ana);
            double nm[4][4], nt[4][4];
            for (int r = 0; r < 4; r++) { nm[r][0] = cam->forward[r]; nm[r][1] = cam->left[r]; nm[r][2] = cam->up[r]; nm[r][3] = ana[r]; }
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) nt[r][c] = nm[c][r];
            double *vecs[3] = {cam->forward, cam->left, cam->up};
            for (double *v : vecs) {
                double t1[4], t2[4];
                mat4_vec(nt, v, t1);
                mat4_vec(rot, t1, t2);
                mat4_vec(nm, t2, v);
            }
            find_orthonormal_4(cam->forward, cam->left, cam->up, ana);
            reorthonormalize_4(cam->forward, cam->left, cam->up, ana);
        }
    }
    const double distance = speed * delta_millis;
    if (distance == 0.0) return EU_OK;
    double direction[4] = {0.0, 0.0, 0.0, 0.0}, ana[4];
    auto add = [&](const double *v, double k) { for (int i = 0; i < 4; i++) direction[i] = k > 0 ? direction[i] + v[i] : direction[i] - v[i]; };
    if (in->keys & EU_KEY_W) add(cam->forward, 1);
    if (in->keys & EU_KEY_S) add(cam->forward, -1);
    if (in->keys & EU_KEY_A) add(cam->left, 1);
    if (in->keys & EU_KEY_D) add(cam->left, -1);
    if (in->keys & EU_KEY_LSHIFT) add(cam->up, 1);
    if (in->keys & EU_KEY_LCONTROL) add(cam->up, -1);
    if (in->keys & EU_KEY_Q) { find_orthonormal_4(cam->forward, cam->left, cam->up, ana); add(ana, 1); }
    if (in->keys & EU_KEY_E) { find_orthonormal_4(cam->forward, cam->left, cam->up, ana); add(ana, -1); }
    return move_camera(cam, 4, direction, distance, trace_path, false);
}

}  // namespace

int camera_update(eu_camera *cam, const eu_input *in, const TracePathFn &trace_path) {
    if (!cam || !in || (cam->dim != 3 && cam->dim != 4)) return EU_ERR_INVALID_ARGUMENT;
    const double sens = in->mouse_sensitivity != 0.0 ? in->mouse_sensitivity : 0.01;
    const double speed = in->speed != 0.0 ? in->speed : 10.0;
    eu_camera work = *cam;                     /* a failed update leaves the pose as it was */
    const int rc = cam->dim == 3 ? update3(&work, in, sens, speed, trace_path) : update4(&work, in, speed, trace_path);
    if (rc == EU_OK) *cam = work;
    return rc;
}

}  // namespace euclider
