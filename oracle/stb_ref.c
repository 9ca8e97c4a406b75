/* oracle/stb_ref.c -- TEST INFRASTRUCTURE, never linked into or loaded by the product.
 * The reference decodes PNG / JPEG files with the single-header library it vendors, external/stb_image.h, through
 * stbi_load_from_memory(bytes, size, &w, &h, &channels, 4) (/root/reference/src/TextureLoader.cpp:225-257). This translation unit compiles
 * THAT header where it lies (-I/root/reference/external, see oracle/Makefile: target _ref/libstb_ref.so) so that the product's own decoders
 * (hobbyrenderer_amd/csrc/host/ImageDecode.cpp) can be pinned byte for byte to the reference's decoder (tests/test_decoders_vs_stb.py). */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"
