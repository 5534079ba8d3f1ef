// register_types.cpp -- GDExtension entry point: registers PtkRenderingDevice (the C ABI, method for method) and
// PtkGodotRenderingDevice (RenderingDevice's own signatures) at the scene level.
#include <gdextension_interface.h>

#include <godot_cpp/core/class_db.hpp>
#include <godot_cpp/core/defs.hpp>
#include <godot_cpp/godot.hpp>

#include "ptk_godot_rd.h"
#include "ptk_rendering_device.h"

using namespace godot;

static void initialize_ptk_module(ModuleInitializationLevel p_level) {
    if (p_level != MODULE_INITIALIZATION_LEVEL_SCENE) return;
    ClassDB::register_class<PtkRenderingDevice>();
    ClassDB::register_class<PtkGodotRenderingDevice>();
}

static void uninitialize_ptk_module(ModuleInitializationLevel p_level) { (void)p_level; }

extern "C" GDExtensionBool GDE_EXPORT ptk_library_init(GDExtensionInterfaceGetProcAddress p_get_proc_address,
                                                       const GDExtensionClassLibraryPtr p_library, GDExtensionInitialization* r_initialization) {
    GDExtensionBinding::InitObject init_obj(p_get_proc_address, p_library, r_initialization);
    init_obj.register_initializer(initialize_ptk_module);
    init_obj.register_terminator(uninitialize_ptk_module);
    init_obj.set_minimum_library_initialization_level(MODULE_INITIALIZATION_LEVEL_SCENE);
    return init_obj.init();
}
