// Internal interface of the persistent fast-generation kernel (csrc/ar_persist.hip).
#pragma once
#include "vqw_common.h"

struct ArPersist;
bool arp_supported(const vqw_ar_weights* w, int batch);
int arp_create(ArPersist** out, const vqw_ar_weights* w, const int* dil, const float* const* gated_w,
               const float* const* gated_b, const float* const* out_w, const float* const* out_b, int batch,
               int channels_per_workgroup);   // 0 = auto (4 where the chip has R/4 CUs), 4 or 8
int arp_reset(ArPersist* h, hipStream_t st);
// ONE launch for n handles (1..8, the same kernel instantiation, n * workgroups <= CUs) generating side by side.
// condenc[i]: L+1 device pointers of handle i ([B][2R][Tz] per layer, then [B][S][Tz] of postprocess1)
int arp_run(ArPersist* const* hs, int n, const float* const* const* condenc, int Tz, int ratio, int n_steps, int mode,
            const float* const* uniforms, float* const* audio, int32_t* const* indices, float* const* probs_last,
            hipStream_t st);
int arp_workgroups(const ArPersist* h);
bool arp_same_launch(const ArPersist* x, const ArPersist* y);
int arp_error(ArPersist* h, hipStream_t st);   // 0 ok, 1 a spin-wait timed out, -1 HIP error
void arp_destroy(ArPersist* h);
