"""The reference's builder surface with its own signatures: functions that create their variables
implicitly under a variable scope, and the classes `Encoder_64`, `Wavenet`, `WavenetDecoder`.

The reference builds a TF-1.x graph: `conv1d_v2(net, filters, kernel_size, ...)` calls `tf.get_variable('kernel')`
inside `tf.variable_scope(...)` (wavenet_ops.py:59-90) and classes wire those calls (Encoder/encoder.py:8-26,
Decoder/decoder.py:6-62, Decoder/WaveNet/wavenet.py:7-172).  Here the same calls run eagerly on GPU tensors
(channels-last [B,T,C] like the reference) through `ops.py` -> libvqwave; a `VariableStore` plays the part of
the TF variable collection: a variable is created on first use with the reference's initialiser (or taken from a
preloaded dict, e.g. `model.named_parameters()`), under the same scope path = the names of SURVEY Appendix B.

Forward only: training runs through `model.VQVAE` (explicit backward schedule); the fused generator is
`generator.FastGenerator`.  The `fast_*` functions keep the reference's queue semantics (wavenet_ops.py:163-267):
`init_ops` / `push_ops` are lists of callables standing in for the TF ops of the same names -- run `push_ops`
once per generated sample after `predictions` was computed, as generate.py:109 does in one `sess.run`.
"""
import contextlib
import json
import math

import torch

from . import ops as O

_STATE = {'store': None, 'scope': []}


class VariableStore:
    """name -> tensor, with tf.get_variable's create-or-reuse behaviour."""

    def __init__(self, variables=None, device='cuda', seed=0):
        self.vars = dict(variables or {})
        self.device = torch.device(device)
        self.gen = torch.Generator().manual_seed(seed)

    def get(self, name, shape, init):
        full = '/'.join(_STATE['scope'] + [name])
        v = self.vars.get(full)
        if v is None:
            v = init(tuple(shape), self.gen).to(self.device)
            self.vars[full] = v
        if tuple(v.shape) != tuple(shape):
            raise ValueError('variable %s has shape %s, the graph asks for %s' % (full, tuple(v.shape), tuple(shape)))
        return v.to(self.device).contiguous()

    @contextlib.contextmanager
    def use(self):
        prev = (_STATE['store'], _STATE['scope'])
        _STATE['store'], _STATE['scope'] = self, []
        try:
            yield self
        finally:
            _STATE['store'], _STATE['scope'] = prev


@contextlib.contextmanager
def variable_scope(name):
    _STATE['scope'].append(name)
    try:
        yield
    finally:
        _STATE['scope'].pop()


def get_variable(name, shape, init):
    if _STATE['store'] is None:
        raise RuntimeError('no VariableStore is active: wrap the call in `with VariableStore(...).use():`')
    return _STATE['store'].get(name, shape, init)


def _unit_scaling(factor=1.0):      # tf.uniform_unit_scaling_initializer
    def init(shape, gen):
        lim = math.sqrt(3.0 / math.prod(shape[:-1])) * factor
        return (torch.rand(shape, generator=gen) * 2 - 1) * lim
    return init


def _glorot(shape, gen):            # Keras default kernel initialiser
    k, cin, cout = shape
    lim = math.sqrt(6.0 / (k * cin + k * cout))
    return (torch.rand(shape, generator=gen) * 2 - 1) * lim


def _zeros(shape, gen):
    return torch.zeros(shape)


def _ones(shape, gen):
    return torch.ones(shape)


# ---------------------------------------------------------------------------------- wavenet_ops.py, training graph
def conv1d_v2(net, filters, kernel_size, padding='CAUSAL', dilations=1, log=False, stride=1, use_bias=True):
    """wavenet_ops.py:59-90 (variables `kernel` [k, Cin, filters], `bias` [filters]; `log` only made TF summaries)."""
    kernel = get_variable('kernel', (kernel_size, net.shape[-1], filters), _unit_scaling(1.0))
    bias = get_variable('bias', (filters,), _zeros) if use_bias else None
    return O.conv1d_v2(net, kernel, bias, padding=padding, dilations=dilations, stride=stride)


def add_condition(net, condition):
    """wavenet_ops.py:93-101."""
    if condition is None:
        return net
    kernel = get_variable('kernel', (1, condition.shape[-1], net.shape[-1]), _unit_scaling(1.0))
    return O.add_condition(net, condition, kernel)


def gated_cnn(net, dilation_filters, kernel_size, dilations, local_condition, global_condition):
    """wavenet_ops.py:104-114.  The reference's own decoder passes global_condition=None (decoder.py:34-36: speakers enter
    through concat); a non-None one is a second add_condition under the scope 'global_condition' (:109-110)."""
    kernel = get_variable('kernel', (kernel_size, net.shape[-1], 2 * dilation_filters), _unit_scaling(1.0))
    bias = get_variable('bias', (2 * dilation_filters,), _zeros)
    ck = gk = None
    if local_condition is not None:
        with variable_scope('local_condition'):
            ck = get_variable('kernel', (1, local_condition.shape[-1], 2 * dilation_filters), _unit_scaling(1.0))
    if global_condition is not None:
        with variable_scope('global_condition'):
            gk = get_variable('kernel', (1, global_condition.shape[-1], 2 * dilation_filters), _unit_scaling(1.0))
    return O.gated_cnn(net, kernel, bias, dilations, local_condition, ck, global_condition, gk)


def residual_stack(net, dilation_filters, kernel_size, dilations, skip_filters, residual_filters, local_condition,
                   global_condition=None):
    """wavenet_ops.py:117-138 -> (skip_connection, residual_connection)."""
    with variable_scope('gated'):
        gated = gated_cnn(net, dilation_filters, kernel_size, dilations, local_condition, global_condition)
    with variable_scope('skip'):
        skip = conv1d_v2(gated, skip_filters, 1)
    with variable_scope('residual'):
        res = conv1d_v2(gated, residual_filters, 1)
    return skip, res


# ---------------------------------------------------------------------------------- wavenet_ops.py, one-sample ops
class FIFOQueue:
    """tf.FIFOQueue(capacity, shapes=(batch, channels)) as a device ring (wavenet_ops.py:181-184)."""

    def __init__(self, capacity, batch_size, state_size, device):
        self.buf = torch.zeros(capacity, batch_size, state_size, device=device)
        self.head = 0

    def init(self):                 # q.enqueue_many(zeros): generate.py:105 runs it once
        self.buf.zero_()
        self.head = 0

    def front(self):                # q.dequeue() as far as the value goes; the slot is reused by the push
        return self.buf[self.head]

    def push(self, value):          # dequeue + enqueue([current]) of one sess.run
        self.buf[self.head].copy_(value)
        self.head = (self.head + 1) % self.buf.shape[0]


def _queues(key, n, make):
    """Queues are graph state in the reference (created once by build_generator); here they live in the store."""
    st = _STATE['store']
    if st is None:
        raise RuntimeError('no VariableStore is active')
    full = '/'.join(_STATE['scope'] + [key])
    qs = st.vars.get(full)
    if qs is None:
        qs = st.vars[full] = [make() for _ in range(n)]
    return qs


def linear(net, filters, use_bias=True):
    """wavenet_ops.py:147-160: [b, Cin] -> [b, filters]."""
    kernel = get_variable('kernel', (1, net.shape[-1], filters), _unit_scaling(1.0))
    bias = get_variable('bias', (filters,), _zeros) if use_bias else None
    return O.linear(net, kernel, bias)


def fast_conv1d(current, filters, kernel_size, dilations, batch_size):
    """wavenet_ops.py:163-195 -> (new_state, init_ops, push_ops)."""
    cin = current.shape[-1]
    kernel = get_variable('kernel', (kernel_size, cin, filters), _unit_scaling(1.0))
    bias = get_variable('bias', (filters,), _zeros)
    qs = _queues('queues', kernel_size - 1, lambda: FIFOQueue(dilations, batch_size, cin, current.device))
    new_state = O.linear(current, kernel[kernel_size - 1:kernel_size], bias)
    init_ops, push_ops = [], []
    for i, q in enumerate(qs, start=1):
        past = q.front().clone()
        init_ops.append(q.init)
        push_ops.append(lambda q=q, v=current: q.push(v))       # the dequeued value feeds the next queue
        current = past
        new_state = new_state + O.linear(past, kernel[kernel_size - i - 1:kernel_size - i])
    return new_state, init_ops, push_ops


def fast_condition(net, condition_t):
    """wavenet_ops.py:198-209."""
    if condition_t is None:
        return net
    return net + linear(condition_t, net.shape[-1], use_bias=False)


def fast_gated_cnn(current, dilation_filters, kernel_size, dilations, batch_size, local_condition_t, global_condition_t):
    """wavenet_ops.py:212-237."""
    net, init_ops, push_ops = fast_conv1d(current, 2 * dilation_filters, kernel_size, dilations, batch_size)
    with variable_scope('local_condition'):
        net = fast_condition(net, local_condition_t)
    with variable_scope('global_condition'):
        net = fast_condition(net, global_condition_t)
    gated = torch.tanh(net[:, :dilation_filters]) * torch.sigmoid(net[:, dilation_filters:])
    return gated, init_ops, push_ops


def fast_residual_stack(current, dilation_filters, kernel_size, dilations, batch_size, local_condition_t,
                        global_condition_t, skip_filters, residual_filters):
    """wavenet_ops.py:240-267 -> (skip [b, S], residual [b, R], init_ops, push_ops)."""
    with variable_scope('gated'):
        gated, init_ops, push_ops = fast_gated_cnn(current, dilation_filters, kernel_size, dilations, batch_size,
                                                   local_condition_t, global_condition_t)
    with variable_scope('skip'):
        skip = linear(gated, skip_filters)
    with variable_scope('residual'):
        res = linear(gated, residual_filters)
    return skip, res, init_ops, push_ops


# ---------------------------------------------------------------------------------- classes
class Encoder_64:
    """Encoder/encoder.py:8-26: 6 x [Conv1D(768, 5, stride 2, 'same', relu) -> BatchNormalization] -> Conv1D(latent, 1)
    -> BatchNormalization; BatchNormalization is called without `training`, i.e. with its moving statistics."""

    def __init__(self, latent_dim, filters=768):
        self.latent_dim, self.filters = latent_dim, filters

    def build(self, net):
        for i in range(7):
            sfx = '' if i == 0 else '_%d' % i
            cout, k = (self.filters, 5) if i < 6 else (self.latent_dim, 1)
            with variable_scope('conv1d' + sfx):
                kernel = get_variable('kernel', (k, net.shape[-1], cout), _glorot)
                bias = get_variable('bias', (cout,), _zeros)
            net = O.keras_conv1d(net, kernel, bias, stride=2 if i < 6 else 1, relu=i < 6)
            with variable_scope('batch_normalization' + sfx):
                gamma, beta = get_variable('gamma', (cout,), _ones), get_variable('beta', (cout,), _zeros)
                mean, var = get_variable('moving_mean', (cout,), _zeros), get_variable('moving_variance', (cout,), _ones)
            net = (net - mean) * (gamma * torch.rsqrt(var + 1e-3)) + beta
        return net


class Wavenet:
    """Decoder/WaveNet/wavenet.py:7-172."""

    def __init__(self, args_file='wavenet_parameters.json'):
        if isinstance(args_file, dict):
            args = dict(args_file)
        else:
            with open(args_file) as fh:
                args = json.load(fh)
        assert len(args['dilation_rates']) == args['num_cycles'] * args['num_cycle_layers']          # wavenet.py:13
        self.args = args
        self.receptive_field = (sum(args['dilation_rates']) * (args['kernel_size'] - 1) + 1
                                + args['preprocess']['kernel_size'] - 1)

    def _layers(self):
        a = self.args
        for i, d in enumerate(a['dilation_rates']):
            yield 'cycle_%d/layer_%d' % (1 + i // a['num_cycle_layers'], 1 + i % a['num_cycle_layers']), d

    def build(self, inputs, local_condition=None, global_condition=None):
        """inputs [b, t, 1] raw audio -> (logits [b*t, 256], labels int32 [b*t])  (wavenet.py:24-100)."""
        a = self.args
        self.labels = O.mu_law_encode(inputs, to_int=True).reshape(-1)
        net = O.mu_law_encode(O.shift_right(inputs))
        with variable_scope('preprocess'):
            net = conv1d_v2(net, a['preprocess']['filters'], a['preprocess']['kernel_size'])
        with variable_scope('skip'):
            skip = conv1d_v2(net, a['skip_filters'], kernel_size=1)
        for scope, d in self._layers():
            with variable_scope(scope):
                s, r = residual_stack(net, a['dilation_filters'], a['kernel_size'], d, a['skip_filters'],
                                      a['residual_filters'], local_condition, global_condition)
            skip, net = skip + s, net + r
        with variable_scope('postprocess1'):
            net = conv1d_v2(torch.relu(skip), a['skip_filters'], kernel_size=1)
            if local_condition is not None:
                with variable_scope('local_condition'):
                    net = add_condition(net, local_condition)
            if global_condition is not None:                                   # wavenet.py:89-91
                with variable_scope('global_condition'):
                    net = add_condition(net, global_condition)
        with variable_scope('postprocess2'):
            net = conv1d_v2(torch.relu(net), a['quantization_channels'], kernel_size=1)
        self.logits = net.reshape(-1, a['quantization_channels'])
        return self.logits, self.labels

    def build_generator(self, input_t, local_condition_t, global_condition_t, batch_size):
        """One sample: input_t [b, 1] in [-1, 1], local_condition_t [b, Cc] -> self.predictions [b, 256]; leaves
        self.init_ops / self.push_ops (wavenet.py:103-172).  Call once per sample, then run push_ops."""
        a = self.args
        init_ops, push_ops = [], []
        with variable_scope('preprocess'):
            current, i_, p_ = fast_conv1d(O.mu_law_encode(input_t), a['preprocess']['filters'], a['preprocess']['kernel_size'],
                                          1, batch_size)
            init_ops += i_; push_ops += p_
        with variable_scope('skip'):
            skip = linear(current, a['skip_filters'])
        for scope, d in self._layers():
            with variable_scope(scope):
                s, r, i_, p_ = fast_residual_stack(current, a['dilation_filters'], a['kernel_size'], d, batch_size,
                                                   local_condition_t, global_condition_t, a['skip_filters'],
                                                   a['residual_filters'])
            skip, current = skip + s, current + r
            init_ops += i_; push_ops += p_
        with variable_scope('postprocess1'):
            net = linear(torch.relu(skip), a['skip_filters'])
            with variable_scope('local_condition'):
                net = fast_condition(net, local_condition_t)
            with variable_scope('global_condition'):                           # wavenet.py:160-162
                net = fast_condition(net, global_condition_t)
        with variable_scope('postprocess2'):
            net = linear(torch.relu(net), a['quantization_channels'])
        self.init_ops, self.push_ops = init_ops, push_ops
        self.predictions = torch.softmax(net, dim=-1)
        return self.predictions


class WavenetDecoder:
    """Decoder/decoder.py:6-62: concat the (already embedded) speaker condition, then Wavenet."""

    def __init__(self, args_file):
        self.args_file = args_file
        self.wavenet = Wavenet(args_file)

    def build(self, x, local_condition, global_condition, is_training=True):
        if global_condition is not None:
            local_condition = O.concat(local_condition, global_condition)
        return self.wavenet.build(inputs=x, local_condition=local_condition, global_condition=None)

    def build_generator(self, local_condition, global_condition):
        """-> the concatenated condition [b, Tz, Cc]; per sample call
        `self.wavenet.build_generator(input_t, local_condition[:, i // ratio], None, batch_size)`."""
        if global_condition is not None:
            local_condition = O.concat(local_condition, global_condition)
        return local_condition
