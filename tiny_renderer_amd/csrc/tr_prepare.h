// tr_prepare.h -- host-side pass preparation (shader.rs:183-279), see tr_prepare.cpp.
#pragma once

#include <stdint.h>

#include "tiny_renderer.h"

namespace tr {

// kind: 0 default_prepare, 1 shadow_pass_prepare_1, 2 shadow_pass_prepare_2.
// Returns TR_OK, TR_E_SINGULAR or TR_E_INVALID.  `u` is updated in place like the
// reference's Buffer (shadow_matrix survives from pass 1 into pass 2).
int prepare_uniforms(int kind, tr_uniforms *u, uint32_t width, uint32_t height, const float light[3],
                     const float from[3], const float at[3], const float up[3]);

// shadow_matrix * i_vpmv (shader.rs:763-764)
void shadow_times_inverse(const tr_uniforms *u, float out[16]);

// The 16 occlusion sample offsets (shader.rs:916-929).  TR_OK or TR_E_SINGULAR.
int occlusion_steps(const tr_uniforms *u, float out[48]);

}  // namespace tr
