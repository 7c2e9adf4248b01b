// Internal (not part of the C ABI): the single-launch sequence GRU of gru_seq.hip, called by the public
// ivln_cma_seq_fwd_f32 / ivln_cma_seq_bwd_f32 dispatchers in nn_ops.hip / train_ops.hip.
#pragma once
#include <stdint.h>
extern "C" {
int ivln_gru_seq_fwd_persistent(const float* gi, const float* h0, int64_t ld_h0, const uint8_t* masks, const float* w_hh,
                                const float* b_hh, float* out, int64_t ldo, float* state_out, int64_t ld_so, int T, int N,
                                float* save_r, float* save_z, float* save_n, float* save_ghn, void* sync_ws, void* stream);
int ivln_gru_seq_bwd_persistent(const float* d_out, int64_t ld_dout, const float* r, const float* z, const float* n,
                                const float* ghn, const float* out, int64_t ld_out, const float* h0, int64_t ld_h0,
                                const uint8_t* masks, const float* whh_t, int T, int N, float* dgi, float* dgh, float* hp,
                                void* sync_ws, void* stream);
}
