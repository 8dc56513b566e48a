"""Oracle model assembly -- TEST INFRASTRUCTURE, never imported by the product.

torch-CPU fp32 restatement of model.py:7-159, Encoder/encoder.py:8-26,
Decoder/decoder.py:12-62, Decoder/WaveNet/wavenet.py:24-172 and the drivers'
loop semantics (train.py:99-122, generate.py:103-113).  Parameters live in a
dict keyed by the reference's TF variable names.
PARITY UNPINNED against real TensorFlow (see oracle/__init__.py).
"""
import json
import math

import numpy as np
import torch

from . import ref_ops as R

DEFAULT_WAVENET = {
    "quantization_channels": 256, "num_cycles": 3, "num_cycle_layers": 10,
    "dilation_rates": [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3,
    "kernel_size": 3, "dilation_filters": 256, "skip_filters": 512,
    "residual_filters": 256, "preprocess": {"kernel_size": 32, "filters": 256},
}
DEFAULT_MODEL = {
    "encoder": "64", "use_vq": True, "speaker_embedding": 64, "k": 512,
    "latent_dim": 64, "beta": 0.25, "encoder_filters": 768,
    "learning_rate_schedule": {"0": 8e-5, "80000": 6e-5, "160000": 4e-5,
                               "240000": 2e-5, "320000": 1e-5, "400000": 8e-6},
}


def layer_scope(i, num_cycle_layers):
    """wavenet.py:64-65."""
    return 'decoder/cycle_%d/layer_%d' % (1 + i // num_cycle_layers, 1 + i % num_cycle_layers)


def bn_scope(i):
    return 'encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)


def conv_scope(i):
    return 'encoder/conv1d' + ('' if i == 0 else '_%d' % i)


# ----------------------------------------------------------------------------
# Initialisation (distributions of the reference's initialisers)
# ----------------------------------------------------------------------------
def _uus(rng, shape, factor):
    """tf.uniform_unit_scaling_initializer(factor): U(+-factor*sqrt(3/prod(shape[:-1])))."""
    fan = int(np.prod(shape[:-1]))
    lim = math.sqrt(3.0 / fan) * factor
    return torch.from_numpy(rng.uniform(-lim, lim, size=shape).astype(np.float32))


def _glorot(rng, shape):
    """Keras glorot_uniform on [k,Cin,Cout]."""
    k, cin, cout = shape
    lim = math.sqrt(6.0 / (k * cin + k * cout))
    return torch.from_numpy(rng.uniform(-lim, lim, size=shape).astype(np.float32))


def init_params(mcfg, wcfg, num_speakers, seed=0, randomize_all=False):
    """All trainable variables + BN moving stats.  `randomize_all` perturbs biases and
    BN parameters away from their (zero / one) initial values so that parity tests
    exercise every term."""
    rng = np.random.RandomState(seed)
    P = {}
    F_enc = mcfg.get("encoder_filters", 768)
    D = mcfg["latent_dim"]
    Cs = mcfg["speaker_embedding"]

    def small(n, base=0.0, s=0.1):
        if randomize_all:
            return torch.from_numpy((base + s * rng.standard_normal(n)).astype(np.float32))
        return torch.full((n,), float(base))

    if Cs > 0:
        P['speaker_embedding'] = _uus(rng, (num_speakers, Cs), 2.0)      # model.py:23-26
    if mcfg.get("encoder", "64") == "Magenta":                           # encoder.py:38-63
        Fm = 128
        P['encoder/preprocess/kernel'] = _uus(rng, (5, 1, Fm), 1.0)
        P['encoder/preprocess/bias'] = small(Fm)
        for i in range(6):
            s = 'encoder/cycle_1/layer_%d' % (i + 1)
            for scope, k in (('dilated', 1), ('gate', 5), ('filter', 5), ('residual', 1)):
                P['%s/%s/kernel' % (s, scope)] = _uus(rng, (k, Fm, Fm), 1.0)
                P['%s/%s/bias' % (s, scope)] = small(Fm)
        P['encoder/postprocess/kernel'] = _uus(rng, (1, Fm, D), 1.0)
        P['encoder/postprocess/bias'] = small(D)
    if mcfg.get("encoder", "64") == "2019":                              # encoder.py:72-98
        shapes = [(3, 13, 768), (3, 768, 768), (4, 768, 768)] + [(3, 768, 768)] * 6 + [(1, 768, D)]
        for i, shp in enumerate(shapes):
            P[conv_scope(i) + '/kernel'] = _glorot(rng, shp)
            P[conv_scope(i) + '/bias'] = small(shp[2])
    cin = 1
    for i in range(6 if mcfg.get("encoder", "64") == "64" else 0):       # encoder.py:14-20
        P[conv_scope(i) + '/kernel'] = _glorot(rng, (5, cin, F_enc))
        P[conv_scope(i) + '/bias'] = small(F_enc)
        P[bn_scope(i) + '/gamma'] = small(F_enc, 1.0)
        P[bn_scope(i) + '/beta'] = small(F_enc)
        P[bn_scope(i) + '/moving_mean'] = torch.zeros(F_enc)
        P[bn_scope(i) + '/moving_variance'] = torch.ones(F_enc)
        cin = F_enc
    if mcfg.get("encoder", "64") == "64":
        P[conv_scope(6) + '/kernel'] = _glorot(rng, (1, F_enc, D))        # encoder.py:21-25
        P[conv_scope(6) + '/bias'] = small(D)
        P[bn_scope(6) + '/gamma'] = small(D, 1.0)
        P[bn_scope(6) + '/beta'] = small(D)
        P[bn_scope(6) + '/moving_mean'] = torch.zeros(D)
        P[bn_scope(6) + '/moving_variance'] = torch.ones(D)
    if mcfg.get("use_vq", True):                                         # model.py:137-138: only built under use_vq
        P['embedding/embedding'] = _uus(rng, (mcfg["k"], D), 1.7)        # model.py:47-49

    Cc = D + (Cs if Cs > 0 else num_speakers)                            # decoder_ops.py:39-43 (Cs = 0: the one-hot itself)
    Rf, Sf, Df = wcfg["residual_filters"], wcfg["skip_filters"], wcfg["dilation_filters"]
    pk, pf = wcfg["preprocess"]["kernel_size"], wcfg["preprocess"]["filters"]
    P['decoder/preprocess/kernel'] = _uus(rng, (pk, 1, pf), 1.0)          # wavenet.py:42-44
    P['decoder/preprocess/bias'] = small(pf)
    P['decoder/skip/kernel'] = _uus(rng, (1, pf, Sf), 1.0)               # wavenet.py:53-54
    P['decoder/skip/bias'] = small(Sf)
    ks = wcfg["kernel_size"]
    for i, _ in enumerate(wcfg["dilation_rates"]):                       # wavenet.py:63-70
        s = layer_scope(i, wcfg["num_cycle_layers"])
        P[s + '/gated/kernel'] = _uus(rng, (ks, Rf, 2 * Df), 1.0)
        P[s + '/gated/bias'] = small(2 * Df)
        P[s + '/gated/local_condition/kernel'] = _uus(rng, (1, Cc, 2 * Df), 1.0)
        P[s + '/skip/kernel'] = _uus(rng, (1, Df, Sf), 1.0)
        P[s + '/skip/bias'] = small(Sf)
        P[s + '/residual/kernel'] = _uus(rng, (1, Df, Rf), 1.0)
        P[s + '/residual/bias'] = small(Rf)
    P['decoder/postprocess1/kernel'] = _uus(rng, (1, Sf, Sf), 1.0)       # wavenet.py:80-82
    P['decoder/postprocess1/bias'] = small(Sf)
    P['decoder/postprocess1/local_condition/kernel'] = _uus(rng, (1, Cc, Sf), 1.0)
    P['decoder/postprocess2/kernel'] = _uus(rng, (1, Sf, wcfg["quantization_channels"]), 1.0)
    P['decoder/postprocess2/bias'] = small(wcfg["quantization_channels"])
    return P


def is_trainable(name):
    return not (name.endswith('moving_mean') or name.endswith('moving_variance'))


# ----------------------------------------------------------------------------
# Forward graph
# ----------------------------------------------------------------------------
def encoder_64(x, P, collect=None):
    """encoder.py:13-26.  x [B,T,1] -> z_e [B,T/64,D].  collect: the relu outputs of the six layers ('enc_relu_<i>')."""
    net = x
    for i in range(6):
        net = R.keras_conv1d(net, P[conv_scope(i) + '/kernel'], P[conv_scope(i) + '/bias'],
                             stride=2, padding='same', relu=True)
        if collect is not None:
            collect['enc_relu_%d' % i] = net
        b = bn_scope(i)
        net = R.batch_norm_inference(net, P[b + '/gamma'], P[b + '/beta'],
                                     P[b + '/moving_mean'], P[b + '/moving_variance'])
    net = R.keras_conv1d(net, P[conv_scope(6) + '/kernel'], P[conv_scope(6) + '/bias'],
                         stride=1, padding='valid')
    b = bn_scope(6)
    return R.batch_norm_inference(net, P[b + '/gamma'], P[b + '/beta'],
                                  P[b + '/moving_mean'], P[b + '/moving_variance'])


def encoder_magenta(x, P):
    """encoder.py:38-63 (Encoder_Magenta).  x [B,T,1] -> [B,T/64,D].  NB: tanh is applied to the
    'gate' scope's output and sigmoid to the 'filter' scope's (encoder.py:59)."""
    net = R.mu_law_encode(R.shift_right(x))
    en = R.conv1d_v2(net, P['encoder/preprocess/kernel'], P['encoder/preprocess/bias'])
    for i, d in enumerate([1, 2, 4, 8, 16, 16]):
        s = 'encoder/cycle_1/layer_%d' % (i + 1)
        dd = R.conv1d_v2(en, P[s + '/dilated/kernel'], P[s + '/dilated/bias'], 1, stride=2)
        g = R.conv1d_v2(dd, P[s + '/gate/kernel'], P[s + '/gate/bias'], d)
        f = R.conv1d_v2(dd, P[s + '/filter/kernel'], P[s + '/filter/bias'], d)
        gated = torch.tanh(g) * torch.sigmoid(f)
        en = dd + R.conv1d_v2(gated, P[s + '/residual/kernel'], P[s + '/residual/bias'])
    return R.conv1d_v2(en, P['encoder/postprocess/kernel'], P['encoder/postprocess/bias'])


def encoder_2019(x, P):
    """encoder.py:72-98 (Encoder_2019) + encoder_ops.py:14-70.  x [B,T,1] -> [B,ceil(ceil(T/160)/2),D]."""
    c = lambda i, net, **kw: R.keras_conv1d(net, P[conv_scope(i) + '/kernel'], P[conv_scope(i) + '/bias'], **kw)  # noqa: E731
    net = R.mfcc(x[:, :, 0])
    net = c(0, net, relu=True)
    net = c(1, net, relu=True) + net
    net = c(2, net, stride=2, relu=True)
    for i in (3, 4):
        net = c(i, net, relu=True) + net
    for i in (5, 6, 7, 8):
        relu = c(i, net, relu=True)
        net = relu + relu                          # encoder.py:91-93: doubles, no residual
    return c(9, net, relu=False)


def vq_distances_np(z, emb):
    """model.py:60-61 direct form.  Summation order FIXED by this build (the TF order is
    unknown): d = 0..D-1 sequentially, each (z-e)*(z-e) and each add rounded to fp32."""
    z = np.asarray(z, dtype=np.float32)
    emb = np.asarray(emb, dtype=np.float32)
    acc = np.zeros((z.shape[0], emb.shape[0]), dtype=np.float32)
    for d in range(z.shape[1]):
        diff = (z[:, None, d] - emb[None, :, d]).astype(np.float32)
        acc = (acc + (diff * diff).astype(np.float32)).astype(np.float32)
    return acc


def discretise(z_e, emb):
    """model.py:57-74 -> (q_z_x int64, e_k, z_q).  argmin: lowest index on ties."""
    shp = z_e.shape
    dist = vq_distances_np(z_e.detach().reshape(-1, shp[-1]).numpy(), emb.detach().numpy())
    q = torch.from_numpy(np.argmin(dist, axis=-1).astype(np.int64)).reshape(shp[:-1])
    e_k = emb[q]
    z_q = z_e + (e_k - z_e).detach()
    return q, e_k, z_q


def wavenet_build(x, local_condition, P, wcfg, collect=None, global_condition=None):
    """wavenet.py:24-100.  x [B,T,1] raw audio; -> logits [B*T,256], labels int32 [B*T].  global_condition [B,Tg,Cg]: the
    second add_condition of wavenet_ops.py:109-110 and wavenet.py:89-91 (None in the reference's own decoder.py:34-36)."""
    labels = R.mu_law_encode(x, to_int=True).reshape(-1)
    inputs = R.mu_law_encode(R.shift_right(x))
    net = R.conv1d_v2(inputs, P['decoder/preprocess/kernel'], P['decoder/preprocess/bias'])
    skip = R.conv1d_v2(net, P['decoder/skip/kernel'], P['decoder/skip/bias'])
    if collect is not None:
        collect['inputs'] = inputs
        collect['net0'] = net
        collect['skip0'] = skip
    Df = wcfg["dilation_filters"]
    for i, d in enumerate(wcfg["dilation_rates"]):
        s = layer_scope(i, wcfg["num_cycle_layers"])
        p = {k[len(s) + 1:]: v for k, v in P.items() if k.startswith(s + '/')}
        skip_out, res_out = R.residual_stack(net, p, Df, d, local_condition, global_condition)
        skip = skip + skip_out
        net = net + res_out
        if collect is not None:
            collect['net_%d' % (i + 1)] = net
    net = torch.relu(skip)
    net = R.conv1d_v2(net, P['decoder/postprocess1/kernel'], P['decoder/postprocess1/bias'])
    net = R.add_condition(net, local_condition, P['decoder/postprocess1/local_condition/kernel'])
    net = R.add_condition(net, global_condition, P.get('decoder/postprocess1/global_condition/kernel'))
    if collect is not None:
        collect['post1_pre'] = net
    net = torch.relu(net)
    net = R.conv1d_v2(net, P['decoder/postprocess2/kernel'], P['decoder/postprocess2/bias'])
    if collect is not None:
        collect['skip_sum'] = skip
    return net.reshape(-1, wcfg["quantization_channels"]), labels


def forward(x, speaker_idx, P, mcfg, wcfg, collect=None):
    """model.py:145-151 (build) up to the losses.  x [B,T,1], speaker_idx int64 [B]."""
    enc = mcfg.get("encoder", "64")
    z_e = encoder_64(x, P, collect) if enc == "64" else {"Magenta": encoder_magenta, "2019": encoder_2019}[enc](x, P)  # model.py:36
    if mcfg["use_vq"]:
        q, e_k, z_q = discretise(z_e, P['embedding/embedding'])         # model.py:57-74
    else:
        q, e_k, z_q = None, z_e, z_e
    if mcfg["speaker_embedding"] > 0:
        h = P['speaker_embedding'][speaker_idx].unsqueeze(1)             # model.py:22-27
    else:                                                                # model.py:19-21: self.h stays the one-hot [B,1,S]
        S = P['decoder/postprocess1/local_condition/kernel'].shape[1] - mcfg["latent_dim"]
        h = torch.nn.functional.one_hot(speaker_idx, S).float().unsqueeze(1)
    local_condition = R.concat(z_q, h)                                   # decoder.py:30-31
    logits, labels = wavenet_build(x, local_condition, P, wcfg, collect)
    ce = torch.nn.functional.cross_entropy(logits, labels.long(), reduction='mean')
    out = {'z_e': z_e, 'q': q, 'e_k': e_k, 'z_q': z_q, 'local_condition': local_condition,
           'logits': logits, 'labels': labels, 'reconstruction_loss': ce}
    loss = ce
    if mcfg["use_vq"]:                                                   # model.py:99-106
        out['vq_loss'] = torch.mean((z_e.detach() - e_k) ** 2)
        out['commitment_loss'] = mcfg["beta"] * torch.mean((z_e - e_k.detach()) ** 2)
        loss = loss + out['vq_loss'] + out['commitment_loss']
    out['loss'] = loss
    return out


# ----------------------------------------------------------------------------
# Optimiser (model.py:109-130)
# ----------------------------------------------------------------------------
def lr_at(schedule, step):
    """model.py:111-114: value of the last key <= step (keys visited in file order)."""
    items = [(int(k), float(v)) for k, v in schedule.items()]
    lr = items[0][1]
    for key, value in items:
        if not (step < key):
            lr = value
    return lr


def adam_ema_step(P, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8, decay=0.999):
    """TF-1.x AdamOptimizer + ExponentialMovingAverage(0.999) (model.py:116-128).
    state: {'t': int, 'm': {}, 'v': {}, 'ema': {}} updated in place (fp32)."""
    state['t'] += 1
    t = state['t']
    lr_t = np.float32(lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    for n, g in grads.items():
        m = state['m'].setdefault(n, torch.zeros_like(P[n]))
        v = state['v'].setdefault(n, torch.zeros_like(P[n]))
        e = state['ema'].setdefault(n, P[n].detach().clone())   # shadow initialised to the variable
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        with torch.no_grad():
            P[n].sub_(float(lr_t) * m / (v.sqrt() + eps))
            e.sub_((1 - decay) * (e - P[n]))


def train_step(x, speaker_idx, P, mcfg, wcfg, state, step, collect=None):
    """One sess.run(train_op) (train.py:104-114): returns losses + grads (for tests).  collect: dict that receives the
    forward pass's intermediate tensors (the inputs of the relus among them: 'skip_sum', 'post1_pre', 'enc_relu_<i>')."""
    for n, p in P.items():
        p.requires_grad_(is_trainable(n))
        p.grad = None
    out = forward(x, speaker_idx, P, mcfg, wcfg, collect)
    out['loss'].backward()
    grads = {n: p.grad.detach().clone() for n, p in P.items() if p.grad is not None}
    for p in P.values():
        p.requires_grad_(False)
    lr = lr_at(mcfg["learning_rate_schedule"], step)
    adam_ema_step(P, grads, state, lr)
    return out, grads


# ----------------------------------------------------------------------------
# Fast generation (wavenet.py:103-172 + generate.py:103-113)
# ----------------------------------------------------------------------------
class FastGenerator:
    def __init__(self, P, wcfg, batch):
        self.P, self.w, self.B = P, wcfg, batch
        self.reset()

    def reset(self):
        """sess.run(wavenet.init_ops): all queues filled with zeros (generate.py:105)."""
        w = self.w
        self.pre = R.FastConvState(w["preprocess"]["kernel_size"], 1, self.B, 1)
        self.layers = [R.FastConvState(w["kernel_size"], d, self.B, w["residual_filters"])
                       for d in w["dilation_rates"]]

    def step(self, input_t, cond_t, global_t=None):
        """input_t [B,1] in [-1,1]; cond_t [B,Cc] -> probabilities [B,256].  global_t [B,Cg]: fast_condition under the scope
        'global_condition' (wavenet_ops.py:232-233, wavenet.py:160-162)."""
        P, w = self.P, self.w
        x = R.mu_law_encode(input_t)                                     # wavenet.py:113
        current = self.pre.step(x, P['decoder/preprocess/kernel'], P['decoder/preprocess/bias'])
        skip = R.linear(current, P['decoder/skip/kernel'], P['decoder/skip/bias'])
        Df = w["dilation_filters"]
        for i, _ in enumerate(w["dilation_rates"]):
            s = layer_scope(i, w["num_cycle_layers"])
            net = self.layers[i].step(current, P[s + '/gated/kernel'], P[s + '/gated/bias'])
            net = net + R.linear(cond_t, P[s + '/gated/local_condition/kernel'])
            if global_t is not None:
                net = net + R.linear(global_t, P[s + '/gated/global_condition/kernel'])
            gated = torch.tanh(net[:, :Df]) * torch.sigmoid(net[:, Df:])
            skip = skip + R.linear(gated, P[s + '/skip/kernel'], P[s + '/skip/bias'])
            current = current + R.linear(gated, P[s + '/residual/kernel'], P[s + '/residual/bias'])
        net = torch.relu(skip)
        net = R.linear(net, P['decoder/postprocess1/kernel'], P['decoder/postprocess1/bias'])
        net = net + R.linear(cond_t, P['decoder/postprocess1/local_condition/kernel'])
        if global_t is not None:
            net = net + R.linear(global_t, P['decoder/postprocess1/global_condition/kernel'])
        net = torch.relu(net)
        net = R.linear(net, P['decoder/postprocess2/kernel'], P['decoder/postprocess2/bias'])
        return torch.softmax(net, dim=-1)


def generate(P, wcfg, encoding, length, mode='greedy', uniforms=None):
    """generate.py:103-113.  encoding [B,Tz,Cc]; returns (indices [B,L], audio [B,L])."""
    B = encoding.shape[0]
    gen = FastGenerator(P, wcfg, B)
    audio = np.zeros([B, 1], dtype=np.float32)
    to_write = np.zeros([B, length], dtype=np.float32)
    idx = np.zeros([B, length], dtype=np.int64)
    ratio = length // encoding.shape[1]
    with torch.no_grad():
        for i in range(length):
            probs = gen.step(torch.from_numpy(audio), encoding[:, i // ratio]).numpy()
            if mode == 'greedy':
                pred, decoded = R.decode_greedy(probs)
            else:
                pred, decoded = R.sample_with_uniforms(probs, uniforms[:, i])
            idx[:, i] = pred
            to_write[:, i] = decoded
            audio = decoded[:, None].astype(np.float32)
    return idx, to_write


def synthetic_batch(B, T, num_speakers, seed):
    """SURVEY.md 8(d) synthetic int16 PCM segments -> x [B,T,1] f32, speakers int64 [B]."""
    g = torch.Generator().manual_seed(seed)
    n = torch.arange(T, dtype=torch.float64)
    f = 80 + (3400 - 80) * torch.rand(B, 4, generator=g, dtype=torch.float64)
    ph = 2 * math.pi * torch.rand(B, 4, generator=g, dtype=torch.float64)
    env = 0.6 + 0.4 * torch.sin(2 * math.pi * n / T * (1 + 3 * torch.rand(B, 1, generator=g, dtype=torch.float64)))
    s = torch.sin(2 * math.pi * f[:, :, None] * n / 16000.0 + ph[:, :, None]).sum(1)
    noise = torch.randn(B, T, generator=g, dtype=torch.float64)
    wav = torch.clamp(0.25 * s * env + 0.02 * noise, -1, 1)
    pcm = torch.round(wav * 32767).to(torch.int16)
    x = ((pcm.to(torch.float32) + 0.5) / 32767.5).unsqueeze(-1)
    spk = torch.randint(0, num_speakers, (B,), generator=g)
    return x, spk, pcm
