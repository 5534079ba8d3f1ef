"""Prints the digests libptk uses to recognise the plugin's two procedural-texture files inside an edited shader
source (ptk_procedural_texture_digest).  usage: python tools/proc_texture_digests.py <plugin checkout>
The values go into path-tracer_amd/csrc/ptk_device.h (PTK_DIGEST_*)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "path-tracer_amd", "csrc", "libptk.so"))
lib.ptk_procedural_texture_digest.restype = C.c_int
lib.ptk_procedural_texture_digest.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
base = os.path.join(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", "shaders", "procedural_textures")
for name in ("checker_board.comp", "checker_board_spatial.comp"):
    text = open(os.path.join(base, name), "rb").read()
    d = C.c_uint64(0)
    rc = lib.ptk_procedural_texture_digest(text, len(text), C.byref(d))
    print(name, rc, "0x%016xull" % d.value)
