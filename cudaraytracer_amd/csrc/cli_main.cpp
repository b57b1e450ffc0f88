// Command-line renderer on top of the C ABI: the reference's main() (src/main.cu:20-57) with
// its argument surface (src/command_parse.cuh:52-180) and buffer::save_passes
// (src/buffer.cuh:165-195).  Uses nothing but include/crt.h, so it doubles as the example of
// what a host program calling the library looks like.
//
//   crt_render -s <scene.yaml> [--no-preview|-np] [-o <dir>] [-f png|jpg|bmp|tga|hdr]
//              [-p <passes>] [--dump-every <n>]
//
// There is no window: where the reference opens its preview (the default, without
// --no-preview) this program runs the same calling pattern, one render() call per sample
// (preview.cuh:294-315), and `--dump-every n` writes the colour pass every n samples to
// <dir>/preview.<fmt> in place of the on-screen update.  The `denoised` pass needs Intel Open
// Image Denoise and is not produced.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "crt.h"

namespace {

[[noreturn]] void error_with_message(const std::string& message) {      // utility.cuh:266-269
    std::cerr << message << std::endl;
    std::exit(1);
}

std::string date_time_string() {                                        // utility.cuh:145-151
    std::time_t t = std::time(nullptr);
    std::tm tm = *std::localtime(&t);
    char buf[32];
    std::strftime(buf, sizeof(buf), "%Y%m%d_%H%M%S", &tm);
    return buf;
}

const char* const kPassNames[7] = {"color", "albedo", "emission", "normal", "depth", "opacity", "denoised"};

// parse_passes (command_parse.cuh:52-87): empty -> every pass; otherwise a comma list of names,
// `all`, and `!name`, which removes a name from what has been selected SO FAR.
std::set<std::string> parse_passes(const std::string& value) {
    const std::set<std::string> valid(kPassNames, kPassNames + 7);
    if (value.empty()) return valid;
    std::set<std::string> selected;
    std::stringstream ss(value);
    std::string token;
    while (std::getline(ss, token, ',')) {
        if (token == "all") {
            selected.insert(valid.begin(), valid.end());
        } else if (!token.empty() && token[0] == '!') {
            const std::string negated = token.substr(1);
            if (!valid.count(negated)) error_with_message("Invalid render pass: " + negated);
            selected.erase(negated);
        } else if (valid.count(token)) {
            selected.insert(token);
        } else {
            error_with_message("Invalid render pass: " + token);
        }
    }
    return selected;
}

struct Args {
    std::string scene_file, output_path, output_format;
    bool preview = true;
    std::set<std::string> passes;
    int dump_every = 0;
};

// command_parser (command_parse.cuh:95-139): an option's value is the token after its first
// occurrence; order does not matter.
Args parse_command_line_args(int argc, char** argv) {
    std::vector<std::string> tokens(argv + 1, argv + argc);
    auto option = [&](const std::string& name) -> std::string {
        auto it = std::find(tokens.begin(), tokens.end(), name);
        if (it != tokens.end() && ++it != tokens.end()) return *it;
        return "";
    };
    auto exists = [&](const std::string& name) { return std::find(tokens.begin(), tokens.end(), name) != tokens.end(); };
    Args a;
    a.scene_file = option("-s");
    if (a.scene_file.empty()) error_with_message("No scene file specified");
    a.preview = !(exists("--no-preview") || exists("-np"));
    a.output_path = option("-o");
    if (a.output_path.empty()) a.output_path = "output/" + date_time_string();
    a.output_format = option("-f");
    if (a.output_format.empty()) a.output_format = "png";
    else if (a.output_format != "png" && a.output_format != "jpg" && a.output_format != "bmp" && a.output_format != "tga" && a.output_format != "hdr")
        error_with_message("Invalid output format: " + a.output_format + ". Supported formats are: png, jpg, bmp, tga, hdr");
    a.passes = parse_passes(option("-p"));
    const std::string dump = option("--dump-every");
    if (!dump.empty()) a.dump_every = std::max(0, std::atoi(dump.c_str()));
    return a;
}

void create_directories(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "." && cur != "..") ::mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur += path[i];
    }
}

void check(int status) {
    if (status != CRT_OK) error_with_message(crt_last_error());
}

void write_pass(const std::string& file, const std::string& fmt, int pass, int w, int h, float gamma, const float* data) {
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    check(crt_normalize_pass(pass, w, h, gamma, data, rgb.data()));
    check(crt_write_image(file.c_str(), fmt.c_str(), w, h, rgb.data()));
}

}  // namespace

int main(int argc, char** argv) {
    const Args args = parse_command_line_args(argc, argv);

    crt_host_scene* host = nullptr;                                     // load_scene, scene.cuh:650-690
    check(crt_scene_load_yaml(args.scene_file.c_str(), nullptr, &host));
    const crt_scene_desc* desc = crt_host_scene_desc(host);
    const int w = desc->render.image_width, h = desc->render.image_height, spp = desc->render.samples_per_pixel;
    const float gamma = desc->render.gamma;
    crt_scene* scene = nullptr;
    check(crt_scene_create(desc, &scene));

    std::cout << "Starting render...\n";
    crt_buffers* buffers = nullptr;                                     // create_buffer, buffer.cuh:344-380
    check(crt_buffers_create(w, h, &buffers));
    const size_t n = (size_t)w * h;
    std::vector<float> color(3 * n), albedo(3 * n), emission(3 * n), normal(3 * n), depth(n), opacity(n);
    float render_duration = 0.0f;
    uint64_t rays = 0;
    if (args.preview) {
        // render_with_preview's render thread (preview.cuh:294-315): one sample per call
        if (args.dump_every > 0) create_directories(args.output_path);
        for (int s = 0; s < spp; s++) {
            float seconds = 0.0f;
            uint64_t r = 0;
            check(crt_render(scene, buffers, 1, s, &seconds, &r));
            render_duration += seconds;
            rays += r;
            if (args.dump_every > 0 && ((s + 1) % args.dump_every == 0 || s + 1 == spp)) {
                check(crt_buffers_download(buffers, color.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
                write_pass(args.output_path + "/preview." + args.output_format, args.output_format, 0, w, h, gamma, color.data());
                std::cout << " - " << (s + 1) << " / " << spp << " samples\n";
            }
        }
    } else {
        check(crt_render(scene, buffers, -1, 0, &render_duration, &rays));   // main.cu:35
    }
    if (render_duration < 0) error_with_message("Render failed");
    std::cout << " - Render completed: " << render_duration << " seconds" << std::endl;
    std::cout << " - " << rays << " rays, " << (render_duration > 0 ? rays / render_duration / 1e6 : 0.0) << " Mrays/s\n";

    // buffer::save_passes, buffer.cuh:165-195
    std::cout << "Saving render passes (";
    for (const std::string& p : args.passes) std::cout << " " << p << " ";
    std::cout << ")...\n";
    create_directories(args.output_path);
    check(crt_buffers_download(buffers, color.data(), albedo.data(), emission.data(), normal.data(), depth.data(), opacity.data()));
    const float* data[6] = {color.data(), albedo.data(), emission.data(), normal.data(), depth.data(), opacity.data()};
    for (int p = 0; p < 6; p++)
        if (args.passes.count(kPassNames[p]))
            write_pass(args.output_path + "/" + kPassNames[p] + "." + args.output_format, args.output_format, p, w, h, gamma, data[p]);
    if (args.passes.count("denoised")) std::cout << " - denoised pass skipped: this build has no denoiser\n";
    std::cout << " - Images saved.\n";

    crt_buffers_destroy(buffers);
    crt_scene_destroy(scene);
    crt_host_scene_destroy(host);
    std::cout << "Done.\n";
    return 0;
}
