// Internal interface of the persistent fast-generation kernel (csrc/ar_persist.hip).
#pragma once
#include "vqw_common.h"

struct ArPersist;
bool arp_supported(const vqw_ar_weights* w, int batch);
int arp_create(ArPersist** out, const vqw_ar_weights* w, const int* dil, const float* const* gated_w,
               const float* const* gated_b, const float* const* out_w, const float* const* out_b, int batch);
int arp_reset(ArPersist* h, hipStream_t st);
// condenc: L+1 device pointers ([B][2R][Tz] per layer, then [B][S][Tz] of postprocess1)
int arp_run(ArPersist* h, const float* const* condenc, int Tz, int ratio, int n_steps, int mode, const float* uniforms,
            float* audio, int32_t* indices, float* probs_last, hipStream_t st);
int arp_error(ArPersist* h, hipStream_t st);   // 0 ok, 1 a spin-wait timed out, -1 HIP error
void arp_destroy(ArPersist* h);
