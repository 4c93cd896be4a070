// tr_texels.h -- layout of a scene's interleaved, tiled texel set (fetch_texels, tr_shaders.h; built by tr_scene.cpp).
#pragma once

#include <vector>

#include "tr_math.h"
#include "tr_types.h"

namespace tr {

// words per texel of closure `fs`'s set
constexpr int packed_words(int fs) { return fs == FS_SPECULAR ? 4 : (fs == FS_NORMAL_MAP || fs == FS_DARBOUX) ? 2 : 1; }
// image behind word `word` of closure `fs`'s set (-1: padding)
constexpr int packed_source(int fs, int word)
{
    return word == 0 ? 0
         : fs == FS_SPECULAR ? (word == 1 ? 1 : word == 2 ? 3 : -1)
         : fs == FS_NORMAL_MAP ? (word == 1 ? 1 : -1)
         : fs == FS_DARBOUX ? (word == 1 ? 2 : -1)
         : -1;
}
constexpr int packed_lbw(int words) { return words == 1 ? 3 : 2; }  // log2 of a block's width / height in texels
constexpr int packed_lbh(int words) { return words == 4 ? 1 : 2; }

// a texel of a two-/four-word set: one aligned load
struct alignas(8) Texel2 {
    uint32_t x, y;
};
struct alignas(16) Texel4 {
    uint32_t x, y, z, w;
};

// index (in texels) of texel (cx, cy) in the tiled order
TR_HD uint32_t packed_index(int words, uint32_t bpr, uint32_t cx, uint32_t cy)
{
    const int lbw = packed_lbw(words), lbh = packed_lbh(words);
    return ((mul24(cy >> lbh, bpr) + (cx >> lbw)) << (lbw + lbh)) | ((cy & ((1u << lbh) - 1u)) << lbw) | (cx & ((1u << lbw) - 1u));
}

// Host: the set of closure `fs` from the four rgba8 images (all w x h); `bpr` receives the blocks per row.
inline std::vector<uint32_t> pack_texels(int fs, const uint32_t *const image[4], uint32_t w, uint32_t h, uint32_t &bpr)
{
    const int K = packed_words(fs), lbw = packed_lbw(K), lbh = packed_lbh(K);
    bpr = (w + (1u << lbw) - 1u) >> lbw;
    const uint32_t rows = (h + (1u << lbh) - 1u) >> lbh;
    std::vector<uint32_t> packed(((size_t)bpr * rows << (lbw + lbh)) * K, 0u);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const size_t at = (size_t)packed_index(K, bpr, x, y) * K;
            for (int word = 0; word < K; word++) {
                const int src = packed_source(fs, word);
                if (src >= 0) packed[at + word] = image[src][(size_t)y * w + x];
            }
        }
    return packed;
}

}  // namespace tr
