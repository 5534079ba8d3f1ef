// ptk_source.h -- what libptk reads out of the compute shader text the plugin compiles.
//
// PTUtils.load_shader (proprietary_dataclasses/pt_utils.gd:140-195) edits shaders/ray_tracer.comp as text before
// PTWorkDispatcher.load_shader hands it to RenderingDevice.shader_compile_spirv_from_source
// (pt_work_dispatcher.gd:436-445): it sets the `order` constant and splices the scene's procedural-texture files
// in, each with its function renamed to _procedural_texture_function<i>.  libptk does not compile GLSL; it
// recovers from that text the arguments of ptk_configure: the order, the three compile-time switches
// (ray_tracer.comp:84,92,96) and, for every spliced function, which built-in it is -- recognised by a digest of
// its source with comments and white space removed.
#pragma once
#include <stdint.h>

#include <string>

#include "ptk_device.h"

namespace ptk_source {

struct Info {
    int order = 0;
    int refraction_mode = 2, use_emission = 0, early_t_break = 0;  // the shader's defaults
    int n_proc = 0;
    uint64_t digest[PTK_MAX_PROC] = {0};
};

// The edited shader text -> Info.  False + message when the text is not that shader (no `order` constant) or a
// spliced function cannot be delimited.
bool parse_shader(const char* source, size_t nbytes, Info& out, std::string& err);

// Digest of one procedural-texture file as it is written on disk (function named procedural_texture): equal to the
// digest parse_shader computes for that file's text once spliced in.  False when the text defines no such function.
bool file_digest(const char* text, size_t nbytes, uint64_t& digest, std::string& err);

// PTK_PROC_* of the two files shipped with the plugin (shaders/procedural_textures/checker_board.comp,
// checker_board_spatial.comp), 0 for any other digest.
int builtin_of_digest(uint64_t digest);

}  // namespace ptk_source
