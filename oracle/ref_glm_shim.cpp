// TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own vendored glm 0.9.5.4
// (/root/reference/external/include/glm, header-only), compiled by `make -C oracle ref` into
// oracle/_ref/libglmref.so.  These are the host-side matrix functions on the hot path:
//   glm::inverse(view)                      createRays' ray basis     (cone_tracing_kernels.cu:161-167)
//   glm::rotate / translate / operator*     RGBDCamera::update        (rgbd_camera.cpp:154-160,172-173)
//   vec4 * mat4                             the pose update (Q17)     (rgbd_camera.cpp:172)
//   glm::lookAt                             the views of the tests / bench
// They pin ora_mat4_* (and, through the oracle, the device code) against the reference's own dependency;
// tests/golden/make_ref_glm_golden.py turns their outputs into committed vectors.  Matrices are 16 floats,
// column-major, exactly glm's memory layout.
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/matrix_inverse.hpp>

#include <cstring>

static glm::mat4 load(const float* m) { glm::mat4 r; memcpy(&r[0][0], m, 64); return r; }
static void store(const glm::mat4& m, float* out) { memcpy(out, &m[0][0], 64); }

extern "C" {
void ref_glm_inverse(const float* m, float* out) { store(glm::inverse(load(m)), out); }
void ref_glm_mul(const float* a, const float* b, float* out) { store(load(a) * load(b), out); }
void ref_glm_translate(const float* m, const float* v, float* out) { store(glm::translate(load(m), glm::vec3(v[0], v[1], v[2])), out); }
void ref_glm_rotate_deg(const float* m, float angle, const float* axis, float* out) {
  store(glm::rotate(load(m), angle, glm::vec3(axis[0], axis[1], axis[2])), out);
}
void ref_glm_look_at(const float* eye, const float* center, const float* up, float* out) {
  store(glm::lookAt(glm::vec3(eye[0], eye[1], eye[2]), glm::vec3(center[0], center[1], center[2]), glm::vec3(up[0], up[1], up[2])), out);
}
void ref_glm_vec4_mul_mat4(const float* v, const float* m, float* out) {  // row vector x matrix
  const glm::vec4 r = glm::vec4(v[0], v[1], v[2], v[3]) * load(m);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void ref_glm_mat4_mul_vec4(const float* m, const float* v, float* out) {
  const glm::vec4 r = load(m) * glm::vec4(v[0], v[1], v[2], v[3]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
// this_trans of rgbd_camera.cpp:154-158 exactly as written there
void ref_icp_update_transform(const float* x, float* out) {
  glm::mat4 this_trans =
      glm::rotate(glm::mat4(1.0f), -x[2] * 180.0f / 3.14159f, glm::vec3(0.0f, 0.0f, 1.0f))
    * glm::rotate(glm::mat4(1.0f), -x[1] * 180.0f / 3.14159f, glm::vec3(0.0f, 1.0f, 0.0f))
    * glm::rotate(glm::mat4(1.0f), -x[0] * 180.0f / 3.14159f, glm::vec3(1.0f, 0.0f, 0.0f))
    * glm::translate(glm::mat4(1.0f), glm::vec3(x[3], x[4], x[5]));
  store(this_trans, out);
}
}
