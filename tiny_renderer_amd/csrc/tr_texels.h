// tr_texels.h -- layout of a scene's interleaved, tiled texel set (fetch_texels, tr_shaders.h; built by tr_scene.cpp).
#pragma once

#include "tr_math.h"
#include "tr_types.h"

namespace tr {

// words per texel of closure `fs`'s set: the colour image alone, or -- closures that read a normal map -- four words:
//   word 0 = the colour image's texel (r | g << 8 | b << 16) with, for the specular closure, the specular map's exponent
//            byte in bits 24..31 (the images' own alpha is never stored);
//   words 1..3 = the normal the closure DECODES from its normal map's texel (util.rs:51-56: channel / 255 - 0.5,
//            normalised) as three f32 -- a pure function of the texel, computed once per texel when the scene is made
//            (pack_texels, tr_shaders.h: the very decode_normal the closures would call) instead of once per fragment:
//            three exact divisions, a square root and three more divisions that the fragment stage no longer issues.
constexpr int packed_words(int fs) { return (fs == FS_SPECULAR || fs == FS_NORMAL_MAP || fs == FS_DARBOUX) ? 4 : 1; }
// the normal map closure `fs` decodes (normal_map; darboux: normal_map_tangent), -1: none
constexpr int packed_normal_source(int fs) { return fs == FS_DARBOUX ? 2 : (fs == FS_SPECULAR || fs == FS_NORMAL_MAP) ? 1 : -1; }
constexpr int packed_lbw(int words) { return words == 1 ? 3 : 2; }  // log2 of a block's width / height in texels:
constexpr int packed_lbh(int words) { return words == 4 ? 1 : 2; }  // 8 x 4 or 4 x 2 texels = 128 bytes

// a texel of a four-word set: one aligned load
struct alignas(16) Texel4 {
    uint32_t x, y, z, w;
};

// index (in texels) of texel (cx, cy) in the tiled order
TR_HD uint32_t packed_index(int words, uint32_t bpr, uint32_t cx, uint32_t cy)
{
    const int lbw = packed_lbw(words), lbh = packed_lbh(words);
    return ((mul24(cy >> lbh, bpr) + (cx >> lbw)) << (lbw + lbh)) | ((cy & ((1u << lbh) - 1u)) << lbw) | (cx & ((1u << lbw) - 1u));
}

}  // namespace tr
