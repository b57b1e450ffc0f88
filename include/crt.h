/*
 * crt.h -- C ABI of the MI355X path-tracing core.
 *
 * This library replaces ONE path of JasperJeuken/CudaRayTracer: everything behind
 *     float render(std::unique_ptr<scene>&, std::unique_ptr<buffer>&, int tx, int ty,
 *                  int samples, int samples_done)                       (src/render.cuh:167)
 * i.e. render_kernel (render.cuh:108), get_ray_info (render.cuh:36) and every device
 * function they call, plus the two hand-over structures either side of it:
 * the scene that the kernel reads (device_scene, src/scene.cuh:26-37) and the six
 * pass arrays it writes (buffer, src/buffer.cuh:321-334).
 *
 * Conventions
 *   - every function returns CRT_OK (0) or a negative crt_status; the text of the last
 *     error on the calling thread is returned by crt_last_error().  Nothing throws or
 *     aborts across this boundary (the reference exits the process, utility.cuh:176-184).
 *   - all pointers in the *_desc structures are caller-owned HOST memory that is copied
 *     at crt_scene_create() time; the caller may free it afterwards.
 *   - plain C types only.  float triples are float[3] in x,y,z order.
 *   - one crt_render() may be in flight per scene at a time; it blocks until the device
 *     has finished, like render.cuh:177.
 */
#ifndef CRT_H
#define CRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum crt_status {
    CRT_OK = 0,
    CRT_ERR_INVALID = -1,   /* bad argument / inconsistent description      */
    CRT_ERR_DEVICE = -2,    /* HIP runtime error, no device                 */
    CRT_ERR_PARSE = -3,     /* scene file / OBJ file error                  */
    CRT_ERR_IO = -4,        /* file could not be read or written            */
    CRT_ERR_UNSUPPORTED = -5
} crt_status;

/* enum values follow the reference's declaration order */
enum { CRT_OBJ_BVH = 0, CRT_OBJ_SPHERE = 1, CRT_OBJ_TRI = 2 };              /* hittable.cuh:16-20 */
enum { CRT_MAT_LAMBERTIAN = 0, CRT_MAT_METAL = 1, CRT_MAT_DIELECTRIC = 2,
       CRT_MAT_DIFFUSE_LIGHT = 3 };                                          /* material.cuh:14-20 */
enum { CRT_TEX_SOLID_COLOR = 0, CRT_TEX_CHECKER = 1, CRT_TEX_IMAGE = 2 };    /* texture.cuh:20-24  */
enum { CRT_CAM_PERSPECTIVE = 0, CRT_CAM_ORTHOGRAPHIC = 1 };                  /* camera.cuh:11-15   */

/* One scene object in object-index order: the fields of `hittable` (hittable.cuh:27-37)
 * followed by the `tri` (hittable.cuh:249-260) or `sphere` (hittable.cuh:136-137) payload. */
typedef struct crt_object {
    int32_t type;             /* CRT_OBJ_SPHERE | CRT_OBJ_TRI                              */
    int32_t mat_idx;
    int32_t normal_idx;       /* normal-map texture index or -1                            */
    int32_t visible;          /* 0: dropped from the BVH (bvh.cuh:70-76)                   */
    int32_t shade_smooth;     /* tri only                                                  */
    float bbox_min[3];        /* padded bounding box as the reference stores it            */
    float bbox_max[3];
    float anchor[3];          /* anchor_point                                              */
    float translation[3];
    float rotation[9];        /* row-major m[r][c]                                         */
    float rotation_inv[9];
    /* tri: point, edge1, edge2.   sphere: center.origin, center.direction, {radius,0,0}   */
    float p0[3], p1[3], p2[3];
    float n0[3], n1[3], n2[3];    /* tri vertex normals (n0 = flat normal unless smooth)   */
    float uv0[2], uv1[2], uv2[2]; /* tri                                                   */
    float tangent[3], bitangent[3]; /* tri, precomputed (hittable.cuh:241-245)             */
} crt_object;

/* material.cuh:107-208 */
typedef struct crt_material {
    int32_t type;
    int32_t tex_idx;          /* lambertian, diffuse_light                                 */
    float albedo[3];          /* metal                                                     */
    float fuzz;               /* metal                                                     */
    float ior;                /* dielectric                                                */
} crt_material;

/* texture.cuh:68-124.  Image pixels are 8-bit, `channels` is 1 or 4 (RGB files are padded
 * to RGBA by the loader, texture.cuh:150-151), rows top to bottom, tightly packed. */
typedef struct crt_texture {
    int32_t type;
    float color[3];           /* solid_color                                               */
    int32_t odd_idx, even_idx;/* checker                                                   */
    float inv_scale;          /* checker                                                   */
    int32_t width, height, channels; /* image                                              */
    const uint8_t* pixels;    /* image, width*height*channels bytes                        */
} crt_texture;

/* Precomputed camera exactly as the host constructors leave it
 * (perspective camera.cuh:121-151, orthographic camera.cuh:204-220). */
typedef struct crt_camera {
    int32_t type;
    float center[3];          /* perspective only                                          */
    float lower_left_corner[3]; /* centre of the UPPER-left pixel (camera.cuh:143-145)     */
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float u[3], v[3], w[3];
    float defocus_disk_u[3], defocus_disk_v[3];
    float defocus_angle;      /* <= 0 : pinhole                                            */
} crt_camera;

/* camera.cuh:21-27 */
typedef struct crt_render_info {
    int32_t image_width, image_height, samples_per_pixel, max_bounces;
    float gamma;
} crt_render_info;

/* Flat BVH4 in the reference's layout (bvh.cuh:17-23): 4 slots per node, slot = node*4+i,
 * root = node 0; empty slot = (+inf,-inf,-1,-1) (bvh.cuh:236-239). */
typedef struct crt_bvh4 {
    int32_t slot_count;       /* 4 * node count (the reference calls this node_count)      */
    const float* min_bounds;  /* [slot_count][3] */
    const float* max_bounds;  /* [slot_count][3] */
    const int32_t* children;  /* [slot_count]    */
    const int32_t* object_index; /* [slot_count] */
} crt_bvh4;

/* What scene::scene() hands to the device (scene.cuh:26-37). */
typedef struct crt_scene_desc {
    int32_t object_count, material_count, texture_count, background_index;
    const crt_object* objects;
    const crt_material* materials;
    const crt_texture* textures;
    crt_bvh4 bvh;
    crt_camera camera;
    crt_render_info render;
} crt_scene_desc;

/* ------------------------------------------------------------------ host scene layer -- */

typedef struct crt_host_scene crt_host_scene;   /* parsed scene, host memory only */

/* Decodes an image file into 8-bit pixels (what stbi_load does at texture.cuh:143-147).
 * Must return 0 and fill width/height/channels(1,3 or 4)/pixels (malloc'ed, the library
 * frees it with free()) or non-zero on failure. */
typedef int (*crt_image_loader_fn)(const char* filename, float hdr_gamma, float hdr_scale,
                                   int desired_channels, int flip_y, int* width, int* height,
                                   int* channels, uint8_t** pixels, void* user);

/* Install the decoder used for `type: image` textures (NULL restores the built-in one). */
int crt_set_image_loader(crt_image_loader_fn fn, void* user);

/* The image-file load of an image texture (image_texture's constructor, texture.cuh:143-158:
 * stbi_hdr_to_ldr_gamma/scale, flip, stbi_load with `desired_channels`, 0 = as stored) through
 * the installed loader or the built-in decoders (PNG, JPEG, Radiance HDR, PNM).  `*pixels` is
 * malloc'ed; release it with crt_free(). */
int crt_load_image(const char* filename, float hdr_gamma, float hdr_scale, int desired_channels, int flip_y,
                   int* width, int* height, int* channels, uint8_t** pixels);
void crt_free(void* p);

/* load_scene() (scene.cuh:650-690): parse a YAML scene file, expand quads/boxes/models,
 * apply translate/rotate, build the BVH4.  `base_dir` (may be NULL) is prepended to
 * relative asset file names. */
int crt_scene_load_yaml(const char* scene_file, const char* base_dir, crt_host_scene** out);
/* Same from a YAML string already in memory. */
int crt_scene_parse_yaml(const char* yaml_text, const char* base_dir, crt_host_scene** out);
/* Override the render settings of a parsed scene (negative / zero = keep). Rebuilds the
 * camera because pixel deltas depend on the image size. */
int crt_host_scene_set_render(crt_host_scene* hs, int width, int height, int samples, int max_bounces);
const crt_scene_desc* crt_host_scene_desc(const crt_host_scene* hs);
void crt_host_scene_destroy(crt_host_scene* hs);

/* create_bvh4() (bvh.cuh:299-310) on its own: n objects given by padded boxes and
 * visibility flags.  Returns the slot count (4*nodes) or a negative status; call with
 * NULL outputs to size the arrays (capacity is in slots). */
int crt_bvh4_build(int n, const float* bbox_min, const float* bbox_max, const int32_t* visible,
                   int capacity_slots, float* min_bounds, float* max_bounds,
                   int32_t* children, int32_t* object_index);

/* -------------------------------------------------------------------- device layer ---- */

typedef struct crt_scene crt_scene;       /* scene resident in HBM of one GPU */
typedef struct crt_buffers crt_buffers;   /* the six pass arrays in HBM       */

int crt_device_count(int* count);
/* Select the GPU the calling process renders on (one process per GPU). */
int crt_set_device(int device);

/* How a scene is prepared and rendered.  Every field has a default (0) that
 * crt_scene_options_init() sets; crt_scene_create() uses the defaults.  Nothing in the
 * library reads the process environment. */
enum { CRT_ACCEL_AUTO = 0,        /* own hierarchy over the reference's leaf boxes; the reference's
                                     tree in the reference's order when the image depends on it   */
       CRT_ACCEL_REFERENCE = 1,   /* walk the reference's BVH4 (bvh.cuh:299-310), nearest child first */
       CRT_ACCEL_ORDERED = 2 };   /* the reference's BVH4 in the order of bvh.cuh:387-419          */
typedef struct crt_scene_options {
    int32_t struct_size;      /* sizeof(crt_scene_options), set by crt_scene_options_init      */
    int32_t accel;            /* CRT_ACCEL_*                                                   */
    int32_t mode;             /* initial geometry access mode, see crt_scene_set_mode          */
    int32_t ieee_division;    /* 1: never use the Newton-refined reciprocal                    */
    /* Tuning and comparison knobs (0 = the library's choice), described in crt_dev.h: */
    int32_t work_items, blocks, ready_percent, prim_batch;
    int32_t deterministic;    /* 1: one sample per chunk, so every pixel sum is added in sample order and
                                 a render is bit-reproducible from run to run (costs 56 B of scratch per
                                 pixel AND sample instead of per pixel and chunk; a call may then render
                                 at most samples_per_pixel samples)                              */
    /* Development and comparison switches (0 = default), described in crt_dev.h: */
    int32_t wavefront, wavefront_paths, flat_cuboids, general_kernel;
    int32_t exact_ties;       /* 1: two hits whose ray parameters differ in the last bits are told apart the way the
                                 reference's ordered walk does it (bvh.cuh:398-411: a candidate counts only if its BOX is
                                 still hit on [t_min, closest so far], which fails for a hit that rounding puts just in
                                 front of its own box): boxes and primitives are culled against the closest hit times
                                 (1 + 2^-12) and the two candidates are compared with both box entry distances.  BVH4
                                 walk only (the small-scene mode keeps no entry distances and is not used); about a
                                 tenth slower.  0 (default): the nearer hit wins, exact ties by visiting rank; differs
                                 from the reference in about one path in 10^8 (a ray through an edge that lies in a
                                 box face)                                                             */
    int32_t item_order;       /* described in crt_dev.h (0 = default)                                   */
    int32_t chip_parts;       /* n > 1: a render of this scene occupies 1/n of the GPU (its persistent kernel is
                                 started with 1/n of the workgroups that fill the chip).  For a caller that keeps n
                                 renders in flight on n scene copies (crt_render_launch on each, then
                                 crt_render_finish): side by side, each render's last long paths overlap the
                                 others' bulk instead of leaving the GPU half empty.  0 or 1 (default): the whole GPU */
} crt_scene_options;
int crt_scene_options_init(crt_scene_options* options);

/* scene::scene() upload (scene.cuh:61-89): validates the description, repacks it into
 * the device layout and copies it to HBM.  All device memory a render of this scene needs
 * (chunk sums, counters, traversal stacks) is allocated here; crt_render allocates nothing,
 * like render() (render.cuh:167-181). */
int crt_scene_create(const crt_scene_desc* desc, crt_scene** out);
int crt_scene_create_ex(const crt_scene_desc* desc, const crt_scene_options* options, crt_scene** out);
void crt_scene_destroy(crt_scene* sc);

/* create_buffer() (buffer.cuh:344-380).  The six arrays are laid out like the reference:
 * color, albedo, emission, normal = float[W*H][3]; depth, opacity = float[W*H];
 * pixel_index = j*W + i, row 0 = top (render.cuh:113).
 * crt_buffers_create allocates them; crt_buffers_wrap uses ONE caller-provided device
 * allocation of 14*W*H floats holding the arrays back to back in the order above
 * (so that a caller can hand the same memory to a collective). */
int crt_buffers_create(int width, int height, crt_buffers** out);
int crt_buffers_wrap(int width, int height, void* device_ptr_14wh_floats, crt_buffers** out);
void crt_buffers_destroy(crt_buffers* buf);
/* Copy passes to host arrays (any pointer may be NULL to skip that pass). */
int crt_buffers_download(const crt_buffers* buf, float* color, float* albedo, float* emission,
                         float* normal, float* depth, float* opacity);
/* Copy host arrays into the device passes (to resume a progressive render). */
int crt_buffers_upload(crt_buffers* buf, const float* color, const float* albedo,
                       const float* emission, const float* normal, const float* depth,
                       const float* opacity);

/* render() (render.cuh:167-181).
 *   samples < 0  -> samples_per_pixel; else min(samples, samples_per_pixel) (render.cuh:120-124)
 *   sample numbers samples_done+1 .. samples_done+samples are folded into the running mean
 *   held in `buf`; if samples_done == 0 the buffers are overwritten (render.cuh:128-152).
 *   seconds_out (may be NULL): device time of the launch(es), like the clock() pair at
 *   render.cuh:173-180.   rays_out (may be NULL): closest-hit queries traced = calls of
 *   bvh4_hit at render.cuh:47.
 * The random numbers of a path depend only on (pixel, sample number), so splitting a
 * render into several calls, or over several GPUs, gives the same image. */
int crt_render(crt_scene* sc, crt_buffers* buf, int samples, int samples_done,
               float* seconds_out, uint64_t* rays_out);

/* crt_render() for callers that cannot allocate device memory at all (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, semantics and kernels as crt_render(), but the chunk sums and
 * counters live in a 64 MiB array that is part of the library's code object (one copy per GPU,
 * held until the process ends) instead of the scratch the scene owns, and the call itself runs no
 * hipMalloc/hipFree.  Limits: one chunk of sums = 56 bytes per pixel of the image padded to 8x8
 * (8x4 in the tree modes) tiles must fit, i.e. at most about 1.19 M pixels (1000x1000 fits,
 * 1280x1024 does not) - larger images get CRT_ERR_UNSUPPORTED; with fewer chunks than crt_render()
 * would use, the launch has fewer work items and balances worse (1000x1000: 1 chunk instead of 4).
 * Every call on a GPU shares that GPU's array, which keeps the last call's contents (counters are cleared at the start of
 * a call): two calls that overlapped on one GPU (two scenes, two threads) would race on it, so a call that finds another
 * one running on its GPU returns CRT_ERR_INVALID.  The work items are always handed out in the natural order.  Sums are added in a different grouping than in crt_render() when the chunk
 * counts differ, so results agree to float summation order, and bit for bit when both calls use one
 * sample per chunk (small images, or crt_scene_options.deterministic). */
int crt_render_static(crt_scene* sc, crt_buffers* buf, int samples, int samples_done,
                      float* seconds_out, uint64_t* rays_out);

/* Multi-GPU form: render sample numbers first_sample+1 .. first_sample+samples and leave
 * their MEAN in `buf` (overwriting it), without the samples_per_pixel clamp.  Ranks then
 * combine means weighted by their sample counts. */
int crt_render_range(crt_scene* sc, crt_buffers* buf, int first_sample, int samples,
                     float* seconds_out, uint64_t* rays_out);

/* Multi-GPU by image tiles instead of sample ranges: the image is cut into bands of `band_rows`
 * pixel rows (a positive multiple of 8) dealt round-robin to `band_count` ranks; rank
 * `band_index` owns the rows j with (j / band_rows) % band_count == band_index and renders ALL
 * samples of those rows.  Its buffers (crt_buffers_create / _wrap with the scene's width and
 * crt_band_row_count() rows) hold just these rows, top to bottom, in the usual six-array layout,
 * so a rank's result is one contiguous block that an all-gather can send as it is; there is no
 * reduction.  Pixels get the same samples as in a single-GPU render (the RNG is keyed by the
 * pixel's position in the whole image).  crt_render_bands leaves the MEAN of sample numbers
 * first_sample+1 .. first_sample+samples in `own` like crt_render_range; the _launch form returns
 * without waiting (finish with crt_render_finish). */
int crt_band_row_count(int height, int band_rows, int band_index, int band_count);
int crt_render_bands(crt_scene* sc, crt_buffers* own, int first_sample, int samples,
                     int band_rows, int band_index, int band_count, float* seconds_out, uint64_t* rays_out);
int crt_render_bands_launch(crt_scene* sc, crt_buffers* own, int first_sample, int samples,
                            int band_rows, int band_index, int band_count);
/* After the all-gather: `gathered` (device memory) holds the band buffers of ranks 0..band_count-1
 * at gathered + r * rank_stride_floats; writes the whole image into `full`.  Runs on `hip_stream`
 * (a hipStream_t, NULL = the default stream) without waiting. */
int crt_buffers_assemble_bands(crt_buffers* full, const float* gathered, size_t rank_stride_floats,
                               int band_rows, int band_count, void* hip_stream);

/* Asynchronous pieces of crt_render_range for callers that time with their own events:
 * launch on the scene's stream without waiting, then finish (sync + read ray counter).
 * crt_scene_stream returns the hipStream_t as void*. */
int crt_render_launch(crt_scene* sc, crt_buffers* buf, int first_sample, int samples, int fold_done);
int crt_render_finish(crt_scene* sc, float* kernel_ms_out, uint64_t* rays_out);
void* crt_scene_stream(crt_scene* sc);
/* HIP-event durations of the last finished render on the scene's stream: the path-tracing
 * kernel(s) and the pass-resolve kernel, in milliseconds. */
int crt_render_timings(crt_scene* sc, float* trace_ms, float* resolve_ms);

/* How the kernels reach the geometry: 0 = automatic (default), 1 = BVH4 walk with the scene in
 * HBM, 2 = BVH4 walk over an LDS copy of the scene (small scenes), 3 = flat box loop with a
 * per-lane candidate mask (at most 64 objects).  A mode the scene does not qualify for falls
 * back to the next one.  All modes produce the same image. */
int crt_scene_set_mode(crt_scene* sc, int mode);

/* Which kernel variant the next render of this scene uses, as text for logs and tests, e.g.
 * "mode=flat slots=18 groups=2 reciprocal=newton stack=8+0 nodes=21 objects=36 order=free".
 * order=reference: a primitive of the scene sticks out of the box of its BVH slot (nested
 * translate/rotate wrappers of the reference do that, hittable.cuh:77-127 against :429), so the
 * image depends on the order of the reference's walk (bvh.cuh:387-419) and the scene is walked
 * in exactly that order; order=free: any order gives the same image and the library picks its own.
 * Writes at most `size` bytes including the terminator; returns the full length. */
int crt_scene_describe(crt_scene* sc, char* text, int size);

/* save_passes() (buffer.cuh:165-195): normalise the passes to 8 bit exactly like
 * normalize_* (buffer.cuh:202-318) from HOST float arrays.  `out` = W*H*3 bytes.
 * pass: 0 color 1 albedo 2 emission 3 normal 4 depth 5 opacity. */
int crt_normalize_pass(int pass, int width, int height, float gamma, const float* data, uint8_t* out);
/* write_image() (image.cuh:100-118) for fmt "png" | "bmp" | "tga" | "hdr" | "ppm". */
int crt_write_image(const char* filename, const char* fmt, int width, int height, const uint8_t* rgb);

const char* crt_last_error(void);
const char* crt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CRT_H */
