// rd_core_harness.cpp -- the Godot-free core of the GDExtension class (gdextension/src/ptk_rd_core.h) driven the way
// PTWorkDispatcher drives a RenderingDevice (proprietary_dataclasses/pt_work_dispatcher.gd:168-190, 408-445, 470-497,
// 642-653): shader from source text, resources created without a place, uniforms, uniform sets, compute list.
//
//   rd_core_harness <fixture.bin> <shader_source.glsl> <image_out.f32>
//
// The fixture is the one c_abi_harness.c reads (byte buffers with their (set, binding), one push constant block);
// the shader source is the text the plugin would hand to shader_compile_spirv_from_source.  Two dispatches are
// recorded (every set bound before each, as the plugin does); exit status 0 = every call behaved, the caller
// compares the image with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <string>
#include <vector>

#include "ptk_rd_core.h"

#define CHECK(cond, ...)                      \
    do {                                      \
        if (!(cond)) {                        \
            std::fprintf(stderr, __VA_ARGS__); \
            std::fprintf(stderr, "\n");       \
            return 1;                         \
        }                                     \
    } while (0)

static std::vector<char> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    CHECK(argc == 4, "usage: %s fixture.bin shader_source.glsl image_out.f32", argv[0]);
    const std::vector<char> fx = slurp(argv[1]), src = slurp(argv[2]);
    CHECK(fx.size() > 176 && std::memcmp(fx.data(), "PTKH", 4) == 0, "not a harness fixture");
    size_t off = 4;
    auto take = [&](void* dst, size_t n) { std::memcpy(dst, fx.data() + off, n); off += n; };
    uint32_t version = 0, n_paths = 0, n_buffers = 0;
    int32_t hdr[8];
    unsigned char push[128];
    take(&version, 4); take(hdr, sizeof hdr); take(push, 128); take(&n_paths, 4);
    for (uint32_t i = 0; i < n_paths; i++) { uint32_t len; take(&len, 4); off += len; }  // (the source text carries the functions)
    take(&n_buffers, 4);

    ptk_rd::Core rd;
    CHECK(rd.open(0), "open: %s", rd.last_error().c_str());
    CHECK(rd.shader_create_from_source("void main() {}", 14) == 0 && !rd.last_error().empty(), "a source without the order line was accepted");
    const uint64_t shader = rd.shader_create_from_source(src.data(), src.size());
    CHECK(shader != 0, "shader_create_from_source: %s", rd.last_error().c_str());
    const uint64_t pipeline = rd.compute_pipeline_create(shader);
    CHECK(pipeline != 0, "compute_pipeline_create: %s", rd.last_error().c_str());

    std::map<int, std::vector<ptk_rd::Uniform>> sets;
    std::vector<uint64_t> rids;
    for (uint32_t i = 0; i < n_buffers; i++) {
        int32_t sb[2];
        uint64_t nbytes;
        take(sb, 8); take(&nbytes, 8);
        const uint64_t buffer = rd.storage_buffer_create((size_t)nbytes, fx.data() + off);
        off += (size_t)nbytes;
        CHECK(buffer != 0, "storage_buffer_create: %s", rd.last_error().c_str());
        rids.push_back(buffer);
        ptk_rd::Uniform u;
        u.type = ptk_rd::UNIFORM_TYPE_STORAGE_BUFFER;
        u.binding = sb[1];
        u.ids.push_back(buffer);
        sets[sb[0]].push_back(u);
    }
    const uint64_t image = rd.texture_create(ptk_rd::DATA_FORMAT_R32G32B32A32_SFLOAT, hdr[0], hdr[1], 1 + 2 + 8 + 64 + 128, nullptr, 0);
    CHECK(image != 0, "texture_create(image): %s", rd.last_error().c_str());
    rids.push_back(image);
    {
        ptk_rd::Uniform u;
        u.type = ptk_rd::UNIFORM_TYPE_IMAGE;
        u.binding = 0;
        u.ids.push_back(image);
        sets[PTK_SET_IMAGE].push_back(u);
    }
    {   // the 512 sampler + texture pairs, every one the 8x8 filler (pt_work_dispatcher.gd:342-345, 350-370)
        unsigned char grid[8 * 8 * 4];
        const unsigned char black[4] = {0, 0, 0, 56}, pink[4] = {56, 16, 56, 56};
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) std::memcpy(grid + (i * 8 + j) * 4, ((i % 2) + (j % 2)) % 2 ? black : pink, 4);
        const uint64_t filler = rd.texture_create(ptk_rd::DATA_FORMAT_R8G8B8A8_UNORM, 8, 8, 1 + 64, grid, sizeof grid);
        CHECK(filler != 0, "texture_create(filler): %s", rd.last_error().c_str());
        rids.push_back(filler);
        const uint64_t sampler = rd.sampler_create();
        ptk_rd::Uniform u;
        u.type = ptk_rd::UNIFORM_TYPE_SAMPLER_WITH_TEXTURE;
        u.binding = 0;
        for (int k = 0; k < 512; k++) { u.ids.push_back(sampler); u.ids.push_back(filler); }
        sets[PTK_SET_TEXTURE].push_back(u);
    }
    std::map<int, uint64_t> set_rids;
    for (auto& kv : sets) set_rids[kv.first] = rd.uniform_set_create(kv.second, kv.first);

    CHECK(rd.compute_list_dispatch(hdr[2], hdr[3], 1) != PTK_OK, "a dispatch outside a compute list was accepted");
    for (int pass = 0; pass < 2; pass++) {
        CHECK(rd.compute_list_begin() == PTK_OK, "compute_list_begin: %s", rd.last_error().c_str());
        CHECK(rd.compute_list_bind_compute_pipeline(pipeline) == PTK_OK, "bind pipeline: %s", rd.last_error().c_str());
        for (auto& kv : set_rids) CHECK(rd.compute_list_bind_uniform_set(kv.second, kv.first) == PTK_OK, "bind set %d: %s", kv.first, rd.last_error().c_str());
        CHECK(rd.compute_list_set_push_constant(push, 128) == PTK_OK, "push constants: %s", rd.last_error().c_str());
        CHECK(rd.compute_list_dispatch(hdr[2], hdr[3], 1) == PTK_OK, "compute_list_dispatch: %s", rd.last_error().c_str());
        rd.compute_list_end();
        rd.submit();
        CHECK(rd.sync() == PTK_OK, "sync: %s", rd.last_error().c_str());
    }
    const size_t n_pixels = (size_t)hdr[0] * (size_t)hdr[1];
    std::vector<float> pixels(n_pixels * 4);
    CHECK(rd.texture_get_data(image, pixels.data(), n_pixels * 16) == PTK_OK, "texture_get_data: %s", rd.last_error().c_str());
    FILE* f = std::fopen(argv[3], "wb");
    CHECK(f && std::fwrite(pixels.data(), 16, n_pixels, f) == n_pixels, "cannot write %s", argv[3]);
    std::fclose(f);
    for (auto& kv : set_rids) CHECK(rd.free_rid(kv.second) == PTK_OK, "free uniform set");
    for (uint64_t r : rids) CHECK(rd.free_rid(r) == PTK_OK, "free_rid: %s", rd.last_error().c_str());
    std::printf("rd_core_harness ok: %dx%d\n", (int)hdr[0], (int)hdr[1]);
    return 0;
}
