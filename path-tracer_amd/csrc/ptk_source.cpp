// ptk_source.cpp -- see ptk_source.h.  Plain host code: no HIP calls.
#include "ptk_source.h"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace ptk_source {
namespace {

// Comments blanked out (positions kept), so that names inside comments are never found.
std::string blank_comments(const char* s, size_t n) {
    std::string out(s, n);
    size_t i = 0;
    while (i < n) {
        if (s[i] == '/' && i + 1 < n && s[i + 1] == '/') {
            while (i < n && s[i] != '\n') out[i++] = ' ';
        } else if (s[i] == '/' && i + 1 < n && s[i + 1] == '*') {
            out[i] = out[i + 1] = ' ';
            i += 2;
            while (i < n && !(s[i] == '*' && i + 1 < n && s[i + 1] == '/')) { if (s[i] != '\n') out[i] = ' '; i++; }
            if (i < n) { out[i] = ' '; if (i + 1 < n) out[i + 1] = ' '; i += 2; }
        } else {
            i++;
        }
    }
    return out;
}

bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

size_t skip_space(const std::string& s, size_t i) {
    while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    return i;
}

// FNV-1a, 64 bit, over the text without white space
uint64_t digest_of(const std::string& s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (char c : s) {
        if (std::isspace((unsigned char)c)) continue;
        h ^= (unsigned char)c;
        h *= 0x100000001b3ull;
    }
    return h;
}

// The definition `name ( vec3 ...` in comment-free text: position of the name, or npos.
size_t find_definition(const std::string& clean, const std::string& name) {
    size_t p = 0;
    while ((p = clean.find(name, p)) != std::string::npos) {
        const size_t after = p + name.size();
        const bool whole = (p == 0 || !ident_char(clean[p - 1])) && (after >= clean.size() || !ident_char(clean[after]));
        if (whole) {
            size_t q = skip_space(clean, after);
            if (q < clean.size() && clean[q] == '(') {
                q = skip_space(clean, q + 1);
                if (clean.compare(q, 4, "vec3") == 0) return p;
            }
        }
        p = after;
    }
    return std::string::npos;
}

// One procedural-texture function with what its file declares in front of it: from behind the last closing brace
// before the definition to the closing brace of the function's body.
bool delimit(const std::string& clean, size_t name_pos, size_t& first, size_t& last, std::string& err) {
    const size_t prev = clean.rfind('}', name_pos);
    first = prev == std::string::npos ? 0 : prev + 1;
    size_t open = clean.find('{', name_pos);
    if (open == std::string::npos) { err = "procedural texture function has no body"; return false; }
    int depth = 0;
    for (size_t i = open; i < clean.size(); i++) {
        if (clean[i] == '{') depth++;
        else if (clean[i] == '}' && --depth == 0) { last = i + 1; return true; }
    }
    err = "procedural texture function body is not closed";
    return false;
}

bool int_after(const std::string& clean, const char* key, bool need_equals, int& value) {
    size_t p = clean.find(key);
    if (p == std::string::npos) return false;
    p = skip_space(clean, p + strlen(key));
    if (need_equals) {
        if (p >= clean.size() || clean[p] != '=') return false;
        p = skip_space(clean, p + 1);
    }
    if (p >= clean.size() || !(std::isdigit((unsigned char)clean[p]) || clean[p] == '-')) return false;
    value = atoi(clean.c_str() + p);
    return true;
}

void replace_all(std::string& s, const std::string& from, const std::string& to) {
    size_t p = 0;
    while ((p = s.find(from, p)) != std::string::npos) { s.replace(p, from.size(), to); p += to.size(); }
}

}  // namespace

bool parse_shader(const char* source, size_t nbytes, Info& out, std::string& err) {
    if (!source || nbytes == 0) { err = "empty shader source"; return false; }
    const std::string clean = blank_comments(source, nbytes);
    out = Info();
    if (!int_after(clean, "const int order", true, out.order)) {
        err = "the source has no `const int order = N;` line: not the path tracer's compute shader";
        return false;
    }
    (void)int_after(clean, "#define use_refraction_tracking", false, out.refraction_mode);
    (void)int_after(clean, "#define USE_EMISSION", false, out.use_emission);
    (void)int_after(clean, "#define use_early_t_break", false, out.early_t_break);
    for (int i = 1;; i++) {
        const std::string name = "_procedural_texture_function" + std::to_string(i);
        const size_t p = find_definition(clean, name);
        if (p == std::string::npos) break;
        if (i > PTK_MAX_PROC) { err = "more than " + std::to_string(PTK_MAX_PROC) + " procedural textures"; return false; }
        size_t a = 0, b = 0;
        if (!delimit(clean, p, a, b, err)) return false;
        std::string seg = clean.substr(a, b - a);
        replace_all(seg, name, "procedural_texture");
        out.digest[i - 1] = digest_of(seg);
        out.n_proc = i;
    }
    return true;
}

bool file_digest(const char* text, size_t nbytes, uint64_t& digest, std::string& err) {
    if (!text || nbytes == 0) { err = "empty procedural texture source"; return false; }
    const std::string clean = blank_comments(text, nbytes);
    const size_t p = find_definition(clean, "procedural_texture");
    if (p == std::string::npos) { err = "no `vec4 procedural_texture(vec3 ...)` function in the text"; return false; }
    size_t a = 0, b = 0;
    if (!delimit(clean, p, a, b, err)) return false;
    digest = digest_of(clean.substr(a, b - a));
    return true;
}

int builtin_of_digest(uint64_t digest) {
    // digests of the two files the plugin ships (tools/proc_texture_digests.py prints them from a checkout)
    if (digest == PTK_DIGEST_CHECKER_BOARD) return PTK_PROC_CHECKER_BOARD;
    if (digest == PTK_DIGEST_CHECKER_BOARD_SPATIAL) return PTK_PROC_CHECKER_BOARD_SPATIAL;
    return 0;
}

}  // namespace ptk_source
