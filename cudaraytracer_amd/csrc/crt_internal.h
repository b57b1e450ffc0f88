// Declarations shared between the translation units of libcrt_hip.so.
#ifndef CRT_INTERNAL_H
#define CRT_INTERNAL_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/crt.h"
#include "../../include/crt_dev.h"
#include "host_box.h"

namespace crt {

// Records the message returned by crt_last_error() on this thread and returns `code`.
int set_error(int code, const std::string& message);
void clear_error();

// BVH4 build over padded boxes; fills the four flat arrays (4 slots per node).
// Returns the slot count.
int build_bvh4(const std::vector<Box>& boxes, const std::vector<uint8_t>& visible, std::vector<float>& min_bounds,
               std::vector<float>& max_bounds, std::vector<int32_t>& children, std::vector<int32_t>& object_index);

// One node of the hierarchy the device walks: four slots, each empty (box +inf/-inf, both
// indices -1), a primitive (object >= 0) or an inner child (child >= 0).
struct AccelNode {
    float lo[4][3], hi[4][3];
    int32_t child[4], object[4];
};
// Surface-area-heuristic hierarchy over the leaf slots (boxes and objects) of a reference BVH4.
void build_accel(const crt_bvh4& ref, std::vector<AccelNode>& nodes, int& depth);
// The reference BVH4 itself in that form.
void accel_from_reference(const crt_bvh4& ref, std::vector<AccelNode>& nodes);

// BMP and TGA (bmp_tga_decode.cpp): what stb_image v2.30 returns for the same file; `ch` = the file's own channel count
bool looks_like_bmp(const std::vector<uint8_t>& file);
bool decode_bmp(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err);
bool looks_like_tga(const std::vector<uint8_t>& file);
bool decode_tga(const std::vector<uint8_t>& file, int& w, int& h, int& ch, std::vector<uint8_t>& out, std::string& err);

// Image decoding for `type: image` textures. Returns true on success; pixels has
// width*height*channels bytes, channels in {1,3,4}.
bool load_image_file(const std::string& path, float hdr_gamma, float hdr_scale, int desired_channels, bool flip_y,
                     int& width, int& height, int& channels, std::vector<uint8_t>& pixels, std::string& error);

}  // namespace crt

// Parsed scene kept on the host.
struct crt_host_scene {
    std::vector<crt_object> objects;
    std::vector<crt_material> materials;
    std::vector<crt_texture> textures;
    std::vector<std::vector<uint8_t>> image_store;   // one entry per texture (empty unless image)
    int background_index = -1;

    // camera_information (camera.cuh:33-41) as parsed
    int cam_type = CRT_CAM_PERSPECTIVE;
    float cam_from[3] = {0, 0, 0}, cam_to[3] = {0, 0, 0}, cam_up[3] = {0, 1, 0};
    float cam_width = 0, cam_vfov = 0, cam_defocus_angle = 0, cam_focus_dist = 10;
    crt_render_info render = {0, 0, 0, 50, 2.0f};
    crt_camera camera = {};

    std::vector<float> bvh_min, bvh_max;
    std::vector<int32_t> bvh_children, bvh_object_index;

    crt_scene_desc desc = {};
    void refresh_desc();
    void rebuild_camera();
};

#endif
