/*
 * c_abi_harness.c -- a plain C99 user of include/ptk.h, with no Python or C++ in between.
 *
 * Plays the part of the plugin's PTWorkDispatcher (create_buffers + create_compute_list,
 * proprietary_dataclasses/pt_work_dispatcher.gd:168-290,470-497) from a fixture file that holds the byte
 * buffers, the (set, binding) they go to, the pipeline switches and one 128-byte push constant block:
 *
 *   c_abi_harness <fixture.bin> <image_out.f32> <display_out.rgba8>
 *
 * It renders one dispatch, writes the rgba32f image and its RGBA8 display form, and exercises two error
 * paths.  Exit status 0 = every call behaved; the caller compares the files with the CPU oracle.
 *
 * Fixture layout (little endian): "PTKH", u32 version = 1, i32 width, height, groups_x, groups_y, order,
 * refraction_mode, use_emission, early_t_break, 128 bytes push constants, u32 n_paths, n_paths x (u32 len,
 * bytes), u32 n_buffers, n_buffers x (i32 set, i32 binding, u64 nbytes, bytes).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptk.h"

static int read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

#define CHECK(cond, ...)                  \
    do {                                  \
        if (!(cond)) {                    \
            fprintf(stderr, __VA_ARGS__); \
            fprintf(stderr, "\n");        \
            return 1;                     \
        }                                 \
    } while (0)

int main(int argc, char** argv) {
    FILE* f;
    char magic[4];
    uint32_t version, n_paths, n_buffers, i;
    int32_t hdr[8];
    unsigned char push[128];
    char* paths[16];
    ptk_ctx* ctx;
    uint64_t image, handles[32];
    float* pixels;
    unsigned char* display;
    size_t n_pixels;
    float ms = 0.0f;

    CHECK(argc == 4, "usage: %s fixture.bin image_out.f32 display_out.rgba8", argv[0]);
    f = fopen(argv[1], "rb");
    CHECK(f != NULL, "cannot open %s", argv[1]);
    CHECK(read_exact(f, magic, 4) == 0 && memcmp(magic, "PTKH", 4) == 0, "not a harness fixture");
    CHECK(read_exact(f, &version, 4) == 0 && version == 1, "unknown fixture version");
    CHECK(read_exact(f, hdr, sizeof hdr) == 0, "short header");
    CHECK(read_exact(f, push, 128) == 0, "short push constants");
    CHECK(read_exact(f, &n_paths, 4) == 0 && n_paths <= 16, "bad texture path count");
    for (i = 0; i < n_paths; i++) {
        uint32_t len;
        CHECK(read_exact(f, &len, 4) == 0 && len < 4096, "bad path length");
        paths[i] = (char*)malloc(len + 1);
        CHECK(paths[i] != NULL && read_exact(f, paths[i], len) == 0, "short path");
        paths[i][len] = 0;
    }

    ctx = ptk_create(0);
    CHECK(ctx != NULL, "ptk_create: %s", ptk_last_error(NULL));
    CHECK(ptk_configure(ctx, hdr[4], hdr[5], hdr[6], hdr[7], (const char* const*)paths, (int)n_paths) == PTK_OK, "ptk_configure: %s",
          ptk_last_error(ctx));

    CHECK(read_exact(f, &n_buffers, 4) == 0 && n_buffers <= 32, "bad buffer count");
    for (i = 0; i < n_buffers; i++) {
        int32_t sb[2];
        uint64_t nbytes;
        void* bytes;
        CHECK(read_exact(f, sb, 8) == 0 && read_exact(f, &nbytes, 8) == 0 && nbytes > 0 && nbytes < ((uint64_t)1 << 30), "bad buffer header");
        bytes = malloc((size_t)nbytes);
        CHECK(bytes != NULL && read_exact(f, bytes, (size_t)nbytes) == 0, "short buffer");
        handles[i] = ptk_buffer_create(ctx, sb[0], sb[1], bytes, (size_t)nbytes);
        CHECK(handles[i] != 0, "ptk_buffer_create(%d,%d): %s", (int)sb[0], (int)sb[1], ptk_last_error(ctx));
        free(bytes); /* the library copied it */
    }
    fclose(f);

    image = ptk_image_create(ctx, hdr[0], hdr[1]);
    CHECK(image != 0, "ptk_image_create: %s", ptk_last_error(ctx));

    /* error paths: bad arguments come back as codes with a message, nothing aborts */
    CHECK(ptk_dispatch(ctx, push, 0, hdr[3], 1) == PTK_E_INVALID && strlen(ptk_last_error(ctx)) > 0, "zero groups were not refused");
    CHECK(ptk_buffer_update(ctx, 0xDEADu, 0, push, 4) == PTK_E_INVALID, "an unknown handle was not refused");

    CHECK(ptk_dispatch(ctx, push, hdr[2], hdr[3], 1) == PTK_OK, "ptk_dispatch: %s", ptk_last_error(ctx));
    CHECK(ptk_sync(ctx) == PTK_OK, "ptk_sync: %s", ptk_last_error(ctx));
    CHECK(ptk_last_dispatch_ms(ctx, &ms) == PTK_OK && ms > 0.0f, "ptk_last_dispatch_ms: %s", ptk_last_error(ctx));

    n_pixels = (size_t)hdr[0] * (size_t)hdr[1];
    pixels = (float*)malloc(n_pixels * 16);
    display = (unsigned char*)malloc(n_pixels * 4);
    CHECK(pixels != NULL && display != NULL, "out of memory");
    CHECK(ptk_image_read(ctx, image, pixels, n_pixels * 16) == PTK_OK, "ptk_image_read: %s", ptk_last_error(ctx));
    CHECK(ptk_image_read_display(ctx, image, display, n_pixels * 4, PTK_DISPLAY_CANVAS) == PTK_OK, "ptk_image_read_display: %s",
          ptk_last_error(ctx));
    CHECK(ptk_image_read(ctx, image, pixels, 16) == PTK_E_INVALID, "a short image read was not refused");

    f = fopen(argv[2], "wb");
    CHECK(f != NULL && fwrite(pixels, 16, n_pixels, f) == n_pixels, "cannot write %s", argv[2]);
    fclose(f);
    f = fopen(argv[3], "wb");
    CHECK(f != NULL && fwrite(display, 4, n_pixels, f) == n_pixels, "cannot write %s", argv[3]);
    fclose(f);

    for (i = 0; i < n_buffers; i++) CHECK(ptk_free(ctx, handles[i]) == PTK_OK, "ptk_free: %s", ptk_last_error(ctx));
    CHECK(ptk_free(ctx, image) == PTK_OK, "ptk_free(image): %s", ptk_last_error(ctx));
    ptk_destroy(ctx);
    for (i = 0; i < n_paths; i++) free(paths[i]);
    free(pixels);
    free(display);
    printf("c_abi_harness ok: %dx%d, dispatch %.3f ms\n", (int)hdr[0], (int)hdr[1], (double)ms);
    return 0;
}
