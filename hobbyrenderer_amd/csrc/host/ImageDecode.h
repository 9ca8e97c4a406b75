// ImageDecode.h -- texture files to decoded texels for the bindless textures the path tracer samples
// (SampleBindlessTextureLevel / SampleBindlessTextureGrad, src/shaders/Bindless.hlsli:118-132).
// The reference decodes PNG/JPG/... with stb_image forced to 4 channels (src/TextureLoader.cpp:215-250, un-vendored) and parses
// DDS itself (:66-213), leaving block-compressed data and mip chains to the GPU's texture units. Here: PNG (all colour types / bit
// depths / Adam7, own inflate), JPEG (baseline, extended sequential and progressive Huffman; stb_image's integer IDCT / upsampling / colour
// conversion), DDS with every format GetFormatFromDDS maps (:66-135): RGBA8 (UNORM / SRGB), BC1-BC5, BC7 (UNORM / SRGB) to 8-bit
// channels, BC6H (UF16 / SF16) and the 16 / 32-bit float formats to float channels, R16G16 / RGBA16 UNORM to float; all mip levels kept.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hobbyrt {

// Decoded texels in one of the HRPT_TEXTURE_FORMAT_* layouts of include/hobbyrt_pt.h (0 RGBA8_UNORM, 1 RGBA8_SRGB, 2 RGBA16_FLOAT,
// 3 RGBA32_FLOAT), all mip levels of the file, level 0 first, tightly packed in `rgba`.
struct Image { uint32_t width = 0, height = 0; std::vector<uint8_t> rgba; uint32_t format = 0, mipCount = 1; };

bool Inflate(const uint8_t* data, size_t n, std::vector<uint8_t>& out, std::string& err, size_t maxOut = (size_t)-1);   // zlib stream (RFC 1950/1951)
bool DecodePNG(const uint8_t* data, size_t n, Image& out, std::string& err);
bool DecodeDDS(const uint8_t* data, size_t n, Image& out, std::string& err);
bool Bc7TablesConsistent();     // self-check of the BC7 partition / anchor tables (every anchor lies in its own subset)
bool DecodeJPEG(const uint8_t* data, size_t n, Image& out, std::string& err);   // baseline / extended sequential / progressive Huffman, 8 bit, 1 or 3 components
// by content: PNG signature, JPEG SOI, "DDS " magic; anything else is an error naming the format when it is recognisable (KTX2, ...)
bool DecodeImage(const uint8_t* data, size_t n, Image& out, std::string& err);
bool LoadImageFile(const std::string& path, Image& out, std::string& err);

} // namespace hobbyrt
