"""Fast autoregressive generation: host-side mirror of Wavenet.build_generator
(wavenet.py:103-172) + the sampling loop of generate.py:103-113, running entirely on the GPU
through vqw_ar_decode_* (ring buffers instead of FIFO queues, on-device sampling)."""
import ctypes as C

import torch

from . import _lib as L


MAX_ROWS = 4    # batch rows of one persistent handle (its LDS budget); larger batches run as several handles
MAX_GROUP = 8   # handles of one launch (vqw_ar_decode_run_group_async)
CU_MARGIN = 8   # CUs left free when handles share a launch: a workgroup of a persistent grid that finds no free CU
#                 keeps its siblings spinning until their timeout (not applied to the whole-chip layout below, which is
#                 chosen only when it fits exactly and is what BASELINE.json configs[3] on one GPU asks for)


def pick_layout(batch, R, cus):
    """(rows per handle, channels per workgroup [0 = the library's choice]) of a batch.  Rows never interact
    (generate.py:40,103-113), so how they are grouped into handles is free.  Measured at the reference widths
    (tools/ar_layouts.py, us per step; profiles/round3_ar_layouts.txt):
        rows   one handle   one-row handles, 8 channels per workgroup (R/8 = 32 workgroups each), ONE launch
         1        65             --
         2       103             66
         4       127             94
         8   129 (2 x 4 rows)   118 (8 x 32 workgroups = every CU of the chip)
    One-row handles are independent pipelines: a handle's step time grows with its rows (every exchange carries B values per
    channel and every dot product is formed B times), and side by side they only share the fabric the exchanges cross.
      * up to MAX_GROUP rows that fit the chip as one-row handles: one row per handle, 8 channels per workgroup;
      * otherwise handles of up to MAX_ROWS rows (more rows per weight byte streamed: the better AGGREGATE rate once the rows
        no longer fit side by side), as many per launch as fit, the rest in further waves.
    VQW_AR_ROWS / VQW_AR_CPB override."""
    import os
    rows = cpb = None
    if os.environ.get('VQW_AR_ROWS'):
        rows = max(1, min(MAX_ROWS, int(os.environ['VQW_AR_ROWS'])))
    if os.environ.get('VQW_AR_CPB') in ('4', '8'):
        cpb = int(os.environ['VQW_AR_CPB'])
    if rows is None and cpb is None and 1 < batch <= MAX_GROUP and R % 8 == 0 and (R // 8) * batch <= cus:
        return 1, 8
    if rows is None:
        n_parts = -(-batch // MAX_ROWS)
        rows = -(-batch // n_parts)
    return rows, (cpb or 0)


class FastGenerator:
    def __init__(self, model, batch):
        """Uses the model's LIVE variables (call model.use_ema_weights() first to mirror
        generate.py:88-90, which restores the EMA shadows).  Rows of the batch never interact
        (generate.py:40,103-113), so a batch is cut into independent handles (pick_layout); as many of them as the
        chip has CUs for (one resident workgroup per CU, R/4 or R/8 workgroups per handle) share ONE launch and
        generate side by side, the rest follow in further waves."""
        self.model, self.B = model, batch
        cus = torch.cuda.get_device_properties(model.dev).multi_processor_count
        rows, cpb = pick_layout(batch, model.R, cus)
        self._parts = [rows] * (batch // rows) + ([batch % rows] if batch % rows else [])
        P = model.P
        nl, R, S = model.L, model.R, model.S
        w = L.ArWeights()
        w.n_layers, w.kernel_size, w.R, w.S, w.Q, w.Cc, w.pre_k = nl, model.ks, R, S, model.Q, model.Cc, model.pre_k
        self._dil = (C.c_int32 * nl)(*model.dil)
        w.dilations = self._dil
        f4 = 4  # bytes per float

        def arr(ptrs):
            a = (C.c_void_p * nl)(*ptrs)
            return a
        self._gw = arr([P['gated_w'][l].data_ptr() for l in range(nl)])
        self._gb = arr([P['gated_b'][l].data_ptr() for l in range(nl)])
        self._cw = arr([P['cond_w'].data_ptr() + l * 2 * R * f4 for l in range(nl)])
        self._ow = arr([P['out_w'][l].data_ptr() for l in range(nl)])
        self._ob = arr([P['out_b'][l].data_ptr() for l in range(nl)])
        w.gated_w, w.gated_b, w.cond_w, w.out_w, w.out_b = self._gw, self._gb, self._cw, self._ow, self._ob
        w.cond_ld, w.out_ld = model.Mall, S + R
        w.pre_w, w.pre_b = P['pre_w'].data_ptr(), P['pre_b'].data_ptr()
        w.skip0_w, w.skip0_b = P['skip0_w'].data_ptr(), P['skip0_b'].data_ptr()
        w.post1_w, w.post1_b = P['post1_w'].data_ptr(), P['post1_b'].data_ptr()
        w.post1_cond_w, w.post1_cond_ld = P['cond_w'].data_ptr() + nl * 2 * R * f4, model.Mall
        w.post2_w, w.post2_b = P['post2_w'].data_ptr(), P['post2_b'].data_ptr()
        self._w = w
        self._hs = []
        for nb in self._parts:
            h = C.c_void_p()
            L.check(L.lib().vqw_ar_decode_create_ex(C.byref(h), C.byref(w), nb, cpb))
            self._hs.append(h)
        # waves of handles that are co-resident by construction: floor((CUs - margin) / workgroups), at most MAX_GROUP
        # (no margin where the whole batch fits the chip exactly: the one-utterance-per-XCD layout)
        nwg = [L.lib().vqw_ar_decode_workgroups(h) for h in self._hs]
        whole = len(set(nwg)) == 1 and nwg[0] > 0 and len(self._hs) <= MAX_GROUP and nwg[0] * len(self._hs) <= cus
        self._waves, i = [], 0
        while i < len(self._hs):
            if nwg[i] <= 0:                      # launch-per-phase path: one handle at a time
                self._waves.append([i]); i += 1
                continue
            cap = max(1, min(MAX_GROUP, (cus - (0 if whole else CU_MARGIN)) // nwg[i]))
            j = i + 1
            while j < len(self._hs) and j - i < cap and nwg[j] == nwg[i] and (self._parts[j] > 1) == (self._parts[i] > 1):
                j += 1
            self._waves.append(list(range(i, j))); i = j

    def reset(self):
        """sess.run(wavenet.init_ops) (generate.py:105)."""
        for h in self._hs:
            L.check(L.lib().vqw_ar_decode_reset(h, L.stream()))

    def generate(self, encoding, n_steps, mode='greedy', uniforms=None, ratio=None, return_probs=False):
        """encoding [B][Cc][Tz] (model.encode); continues from the current queue state.
        Returns (audio [B][n] float32, indices [B][n] int32[, probs of the last step [B][Q]])."""
        if mode not in ('greedy', 'sample'):
            raise NotImplementedError('decode mode %s not implemented' % mode)   # utils.py:46
        B, Cc, Tz = encoding.shape
        if B != self.B or Cc != self.model.Cc:
            raise ValueError('encoding must be [%d][%d][Tz]' % (self.B, self.model.Cc))
        L.require_cuda(encoding, uniforms)
        ratio = ratio or 64
        dev = encoding.device
        audio = torch.empty(B, n_steps, device=dev)
        idx = torch.empty(B, n_steps, dtype=torch.int32, device=dev)
        probs = torch.empty(B, self.model.Q, device=dev) if return_probs else None
        if mode == 'sample':
            if uniforms is None:
                uniforms = torch.rand(B, n_steps, device=dev)        # np.random.rand in utils.py:22
            if uniforms.shape != (B, n_steps) or uniforms.dtype != torch.float32:
                raise ValueError('uniforms must be float32 [B][n_steps]')
        encoding = encoding.contiguous()
        starts = [sum(self._parts[:i]) for i in range(len(self._parts))]
        rows = [slice(b0, b0 + nb) for b0, nb in zip(starts, self._parts)]
        vp = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
        for wave in self._waves:                 # one launch per wave; the waves follow each other
            hs = (C.c_void_p * len(wave))(*[self._hs[i].value for i in wave])
            L.check(L.lib().vqw_ar_decode_run_group_async(
                hs, len(wave), vp([encoding[rows[i]] for i in wave]), Tz, ratio, n_steps, 0 if mode == 'greedy' else 1,
                vp([uniforms[rows[i]] for i in wave]) if uniforms is not None else None,
                vp([audio[rows[i]] for i in wave]), vp([idx[rows[i]] for i in wave]),
                vp([probs[rows[i]] for i in wave]) if probs is not None else None, L.stream())
                if L.lib().vqw_ar_decode_workgroups(self._hs[wave[0]]) > 0 else
                L.lib().vqw_ar_decode_run_async(
                    self._hs[wave[0]], L.ptr(encoding[rows[wave[0]]]), Tz, ratio, n_steps, 0 if mode == 'greedy' else 1,
                    L.ptr(uniforms[rows[wave[0]]]) if uniforms is not None else None, L.ptr(audio[rows[wave[0]]]),
                    L.ptr(idx[rows[wave[0]]]), L.ptr(probs[rows[wave[0]]]) if probs is not None else None, L.stream()))
            for i in wave:
                L.check(L.lib().vqw_ar_decode_wait(self._hs[i]))
        return (audio, idx, probs) if return_probs else (audio, idx)

    def close(self):
        for h in getattr(self, '_hs', []):
            L.lib().vqw_ar_decode_destroy(h)
        self._hs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
