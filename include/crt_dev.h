/*
 * crt_dev.h -- development side of the C ABI of the MI355X path-tracing core: probes that the
 * known-answer and parity tests call, counters of the profiling build, and the description of the
 * tuning and comparison fields of crt_scene_options.  Nothing here has a counterpart in the reference
 * (JasperJeuken/CudaRayTracer); a program that replaces render() (src/render.cuh:167) needs crt.h only.
 * Same library, same conventions as crt.h.
 */
#ifndef CRT_DEV_H
#define CRT_DEV_H

#include "crt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fields of crt_scene_options (crt.h) for tuning and comparison; 0 = the library's choice.
 *   work_items      work items (pixel tile x sample chunk) a frame is cut into; 0 = 98304, but no chunk under eight
 *                   samples where the launch has that many (a value given here is taken as it is)
 *   blocks          workgroups of the persistent render kernel; 0 = fill the GPU
 *   ready_percent   tree walk: shade once this share of a wave's rays is done; 0 = 65 (45 for hierarchies beyond 4 MB)
 *   prim_batch      tree walk: lanes with leaf candidates that start a test round; 0 = 8 (2 or 16 for hierarchies beyond 4 MB)
 *   wavefront       BVH walk in HBM: 1 = wavefront pipeline (path state in HBM, one trace and one shade kernel per
 *                   bounce, per-path result records: bulk renders reproducible bit for bit; measured slower than the
 *                   fused kernel), 0 or 2 = the fused kernel (default)
 *   wavefront_paths path slots (224 bytes each) of the wavefront buffers; 0 = 67 108 864
 *   flat_cuboids    small scenes: 0 or 1 = slots whose boxes share planes like the faces of an axis-aligned box (a box
 *                   object, the walls of a room) are box-tested together, and the far faces of such a box are tested
 *                   after the ray's other candidates and only where the closest hit is not already in front of them
 *                   (default; same hits, fewer operations), 2 = every slot on its own, 3 = tested together, but the
 *                   far faces like any other candidate (for comparison)
 *   general_kernel  1: always run the kernel variant compiled for every scene feature (spheres, textures, metal and
 *                   glass, camera kinds ...) instead of the one compiled for just what the scene contains; same results
 *   item_order      the order in which a frame's work items (pixel tile x sample chunk) are handed to the waves.
 *                   By cost: most expensive first, by the time each took in the last launch of the same frame (same
 *                   tiles, chunks and bands); the launch then ends on its cheapest items and the GPU runs dry in a
 *                   fraction of the time.  Natural: tile after tile and chunk after chunk.  Which samples an item
 *                   holds does not change, so the sums are the same whatever the order.
 *                   0 = measured: the first six launches of a frame alternate between the natural order and the order
 *                   by cost, the last four of them are timed, and from the seventh on the launches use the order of
 *                   the fastest timed one (by cost only if by at least 0.5 %); 1 = always natural; 2 = by cost from
 *                   the second launch on.  (Deterministic renders and crt_render_static always use the natural order.) */

/* Development aid: loop-occupancy counters of a library built with -DCRT_PROFILE
 * (128 values: site k < 8 -> [2k] wave-level executions, [2k+1] active lanes; flat mode: [16+n] rays
 * with n candidate slots (7 = seven or more), [24+n] waves whose fullest lane held n, [64+b] how
 * often candidate bit b was set in a lane with three or more; [32+s] shader clocks the waves spent
 * in section s of the path loop).  The product build returns CRT_ERR_UNSUPPORTED. */
int crt_profile_counters(uint64_t* out128, int reset);

/* Primary-ray probe used by the parity tests: for every pixel trace the camera ray of
 * sample number `sample` (1-based, as in render.cuh:128) and return the closest hit of
 * bounce 0: object index (-1 = miss), ray parameter t, surface u, v, front_face.
 * Host output arrays of W*H entries each (any may be NULL). */
int crt_trace_primary(crt_scene* sc, int sample, int32_t* object_index, float* t, float* u,
                      float* v, int32_t* front_face);

/* Single-function probes used by the known-answer tests: `count` cases of ONE building block of
 * the path, evaluated on the device by the functions the render kernels call.  Host arrays;
 * integers travel as float bit patterns.
 *   CRT_PROBE_BOX      aabb::hit (aabb.cuh:132-156)  in[14] = min[3] max[3] origin[3] dir[3] tmin tmax
 *                      out[1]: bit 0 = the tree walk's form of the test, bit 1 = the flat mode's (tmax ignored)
 *   CRT_PROBE_OBJECT   hittable::hit (hittable.cuh:426-446)  in[10] = object origin[3] dir[3] time tmin tmax
 *                      out[11] = hit t u v front_face normal[3] p[3]   (normal after the normal map)
 *   CRT_PROBE_TEXTURE  _texture::value (texture.cuh:313-324)  in[6] = texture u v p[3]   out[3] = colour
 *   CRT_PROBE_CAMERA   camera::get_ray (camera.cuh:269-278)   in[3] = i j sample         out[7] = origin[3] dir[3] time */
enum { CRT_PROBE_BOX = 0, CRT_PROBE_OBJECT = 1, CRT_PROBE_TEXTURE = 2, CRT_PROBE_CAMERA = 3 };
int crt_probe(crt_scene* sc, int kind, int count, const float* in, float* out);

#ifdef __cplusplus
}
#endif
#endif /* CRT_DEV_H */
