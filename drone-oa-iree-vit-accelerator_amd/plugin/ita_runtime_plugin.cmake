# ita_runtime_plugin.cmake -- MI355X counterpart of the reference's iree_runtime_plugin.cmake
# (reference iree_runtime_plugin.cmake:1-4, samples/CMakeLists.txt:1-5).
#
# The reference hooks itself into an IREE source build with
#     -DIREE_CMAKE_PLUGIN_PATHS=<repo>          (IREE then includes <repo>/iree_runtime_plugin.cmake)
# and builds a freestanding object (ITA_dispatch_<arch>.o) that iree-compile links into the .vmfb.
# A GPU engine cannot live in a freestanding object, so this plugin builds
#     libita_mi355x.so   (hipcc, --offload-arch=gfx950; exports ITASelfAttention_workgroup & co.)
# and a host-side import shim target the runtime application links; see INTEGRATION.md.
#
# Usage from an IREE build tree:   cmake -DIREE_CMAKE_PLUGIN_PATHS=<this repo>/drone-oa-iree-vit-accelerator_amd/plugin ...
# Stand-alone:                     cmake -S drone-oa-iree-vit-accelerator_amd/plugin -B build && cmake --build build

set(ITA_MI355X_SOURCE_DIR "${CMAKE_CURRENT_LIST_DIR}/..")
add_subdirectory(${CMAKE_CURRENT_LIST_DIR} ita-mi355x-plugin)
