// ImageDecode.h -- texture files to RGBA8_UNORM, mip 0, as the bindless RGBA8 textures the path tracer samples
// (SampleBindlessTextureLevel, src/shaders/Bindless.hlsli:118-123, always lod 0 on this path).
// The reference decodes PNG/JPG/... with stb_image forced to 4 channels (src/TextureLoader.cpp:215-250, un-vendored) and parses
// DDS itself (:71-213), leaving block-compressed data to the GPU's texture units. Here: PNG (all colour types / bit depths /
// Adam7, own inflate), JPEG (baseline / extended sequential, stb_image's integer IDCT / upsampling / colour conversion; progressive files are
// rejected), DDS with RGBA8 or BC1/BC2/BC3/BC4/BC5 payload decoded on the host.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hobbyrt {

struct Image { uint32_t width = 0, height = 0; std::vector<uint8_t> rgba; };

bool Inflate(const uint8_t* data, size_t n, std::vector<uint8_t>& out, std::string& err);   // zlib stream (RFC 1950/1951)
bool DecodePNG(const uint8_t* data, size_t n, Image& out, std::string& err);
bool DecodeDDS(const uint8_t* data, size_t n, Image& out, std::string& err);
bool DecodeJPEG(const uint8_t* data, size_t n, Image& out, std::string& err);   // baseline / extended sequential Huffman, 8 bit, 1 or 3 components
// by content: PNG signature, JPEG SOI, "DDS " magic; anything else is an error naming the format when it is recognisable (KTX2, ...)
bool DecodeImage(const uint8_t* data, size_t n, Image& out, std::string& err);
bool LoadImageFile(const std::string& path, Image& out, std::string& err);

} // namespace hobbyrt
