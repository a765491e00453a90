/* raytrace_oracle.h — TEST INFRASTRUCTURE. C ABI of the scalar CPU
 * restatement of shaders/raytrace.comp (see raytrace_oracle.cpp). */
#ifndef RAYTRACE_ORACLE_H
#define RAYTRACE_ORACLE_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleCounters {
    /* every calculateIntersections call the shader makes (4 per diffuse segment) */
    uint64_t boxTestsReference, triTestsReference, raysReference, raysHitReference;
    /* the subset the wavefront pipeline executes: main rays, plus the NEE ray and
     * the cosine probe of a diffuse bounce when the path continues (same
     * definition as RtCounters.boxTests/triTests/raysTraced/raysHit) */
    uint64_t boxTests, triTests, raysTraced, raysHit;
    uint64_t paths, segments;
    uint64_t stackOverflow; /* traversals that needed more than the shader's 64-entry stack */
    uint64_t emitterTests;  /* emissive primitives the pipeline's light queries test directly (RtCounters.emitterTests) */
    uint64_t lightQueryMismatch; /* light queries whose shortcut answer differs from the shader's closest hit: must be 0 */
} OracleCounters;

/* One dispatch of raytrace.comp main() over rows row0 + k*rowStride.
 * rgba: nRows*width*4 floats, read when progressive, written always. */
int oracle_render(const RtSceneArrays* scene, const PushConstants* pc, uint32_t width, uint32_t height, uint32_t row0,
                  uint32_t rowStride, uint32_t nRows, float* rgba, OracleCounters* counters, int threads);
/* calculateIntersections (raytrace.comp:276-353) per ray */
int oracle_trace_rays(const RtSceneArrays* scene, uint32_t sphereCount, uint32_t objectCount, uint32_t n,
                      const float* origins, const float* dirs, RtHit* out);
void oracle_set_light_queries(int on);
void oracle_set_camera_reuse(int on);
void oracle_set_textures(const RtTexture* textures, uint32_t n);
float oracle_random(uint32_t* state);
void oracle_math_probe(float x, float y, float out[8]);
void oracle_mat4_inverse(const float m[16], float out[16]);
/* include/rt_probe.h on n inputs of 32 floats; 64 floats out each */
void oracle_glsl_probe(uint32_t n, const float* in, float* out);
uint32_t oracle_selftest(void);
unsigned oracle_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
