"""Deferred execution of the MinkowskiEngine call sequence: the reference's OWN code gets the engine's fused launches.

The reference writes a residual block as separate ME calls (/root/reference/models/modules/resnet_block.py:41-57)

    out = self.conv1(x); out = self.norm1(out); out = self.relu(out)          # MinkowskiReLU(inplace=True)
    out = self.conv2(out); out = self.norm2(out)
    if self.downsample is not None: residual = self.downsample(x)
    out += residual; out = self.relu(out)

and the U-Net trunk as conv -> norm -> relu chains joined by `me.cat(out, skip)` (models/res16unet.py:196-270).  Executed
call by call that is an unfused BatchNorm, an elementwise ReLU, an elementwise add and a concat copy per line.  The engine
has ONE kernel for norm (+ residual) (+ ReLU) that can also write straight into a concat buffer, and one engine call per
residual block and direction (csrc/lgs_block.hip) -- so the ME surface RECORDS instead of executing:

  * MinkowskiConvolution[Transpose](x), MinkowskiBatchNorm(x) and ME.cat(...) append an operation to the coordinate
    manager's queue and return a SparseTensor whose features are pending (coordinate maps and kernel maps ARE requested at
    record time: they are built on the engine's map stream while the rest of the forward is still being recorded);
  * MinkowskiReLU(inplace=True) on the pending result of a norm, and `out += residual` on it, are IN-PLACE operations on a
    value nobody has seen yet: they become the norm's epilogue flags (the `+=` also moves the norm behind the producer of its
    residual -- program order of an in-place update);
  * reading `.F` (or shape / dtype / ... ) of any pending tensor executes the whole queue in program order.  The executor
    looks at what is in front of it: [conv3, norm+relu, conv3, (conv1, norm,) norm+residual(+relu)] runs as the whole-block
    autograd node (me/block.py: lgs_block_forward / lgs_block_backward), a norm whose result meets a later `me.cat` writes
    into the shared concat buffer (zero-copy cat), everything else runs module by module with the epilogue flags it collected.

Nothing is re-ordered except the in-place `+=` above, BatchNorm running statistics are updated exactly once per call, module
forward hooks fire at call time as torch defines them (a hook that reads `.F` simply executes what was recorded so far, i.e.
hooks force the call-by-call sequence), `torch.no_grad()` / `enable_grad()` are honoured per recorded call, and switching a
norm between train() and eval() executes what is pending first.  `LGS_DEFER=0` executes every call immediately (the unfused
sequence: same results, tests/test_gpu_reference_calls.py).
"""
import weakref

import torch

from .. import tuning as _tuning

ENABLED = _tuning.host("DEFER") != 0

CONV, BN, CAT = 0, 1, 2
_LIVE = weakref.WeakSet()       # managers with a non-empty queue
UNIT_HOOKS = []                 # [(pre(mod, args, kwargs), post(mod, args, kwargs, out))]: test instrumentation of the executor's
#                                 module-by-module path (the teacher-forced parity tests); their presence disables whole-block nodes
STATS = {"flushes": 0, "ops": 0, "blocks": 0, "cat_hints": 0}


class Op:
    __slots__ = ("kind", "mod", "inp", "out", "relu", "residual", "cat_up", "cat_into", "grad", "uses", "aux")

    def __init__(self, kind, mod, inp, out, aux=None):
        self.kind, self.mod, self.inp, self.out, self.aux = kind, mod, inp, out, aux
        self.relu, self.residual, self.cat_up, self.cat_into = False, None, 0, None
        self.grad = torch.is_grad_enabled()
        self.uses = 0            # recorded consumers of `out`


class _Dead:
    """what `_op` of a tensor becomes when its value will never exist: the queue failed, or the tensor was an intermediate of a
    block that ran as one node"""
    kind = -1

    def __init__(self, why, cause=None):
        self.why, self.cause = why, cause


def _use(t):
    o = t._op
    if o is not None and o.kind >= 0:
        o.uses += 1


def _push(mgr, op):
    q = mgr._pending
    if not q:
        _LIVE.add(mgr)
    q.append(op)


# ------------------------------------------------------------------------------------------------ recording
def record_conv(mod, inp, resolved):
    from .core import SparseTensor
    mgr = inp._manager
    meta = inp._meta if inp._op is not None else (None, inp._F.dtype, inp._F.device)
    out = SparseTensor._pending(resolved[0], mgr, (mod.out_channels, meta[1], meta[2]))
    op = out._op = Op(CONV, mod, inp, out, resolved)
    _use(inp)
    _push(mgr, op)
    return out


def record_bn(mod, inp):
    from .core import SparseTensor
    mgr = inp._manager
    meta = inp._meta if inp._op is not None else (inp._F.shape[1], inp._F.dtype, inp._F.device)
    out = SparseTensor._pending(inp.coordinate_map_key, mgr, meta)
    op = out._op = Op(BN, mod, inp, out)
    _use(inp)
    _push(mgr, op)
    return out


def record_cat(tensors):
    from .core import SparseTensor
    first = tensors[0]
    mgr = first._manager
    ch = sum(t._nch() for t in tensors)
    meta = first._meta if first._op is not None else (None, first._F.dtype, first._F.device)
    out = SparseTensor._pending(first.coordinate_map_key, mgr, (ch, meta[1], meta[2]))
    op = out._op = Op(CAT, None, tuple(tensors), out)
    for t in tensors:
        _use(t)
    _push(mgr, op)
    return out


def relu_inplace(t):
    """MinkowskiReLU(inplace=True) on a pending norm result nobody consumed yet: the norm's ReLU flag.  -> True if absorbed"""
    op = t._op
    if op.kind == BN and op.uses == 0:
        op.relu = True
        return True
    return False


def add_residual(t, other):
    """`t += other` on a pending norm result nobody consumed yet (and not yet rectified): the norm's residual operand; the norm
    moves to the end of the queue, behind the producer of `other`.  -> True if absorbed"""
    op = t._op
    if not (ENABLED and op.kind == BN and op.uses == 0 and not op.relu and op.residual is None):
        return False
    if other is t:
        return False
    op.residual = other
    _use(other)
    q = t._manager._pending
    if q[-1] is not op:
        for i in range(len(q) - 1, -1, -1):
            if q[i] is op:
                del q[i]
                break
        q.append(op)
    return True


# ------------------------------------------------------------------------------------------------ execution
def materialise(t):
    op = t._op
    if op.kind < 0:
        if op.cause is not None:
            raise RuntimeError(op.why) from op.cause
        raise RuntimeError(op.why)
    flush(t._manager)
    if t._op is not None:
        materialise(t)          # dead after the flush: raises


def flush(mgr):
    q = mgr._pending
    if not q:
        return
    mgr._pending = []
    _LIVE.discard(mgr)
    STATS["flushes"] += 1
    STATS["ops"] += len(q)
    try:
        _run(q)
    except BaseException as e:
        for op in q:
            if op.out._op is not None and op.out._op.kind >= 0:
                op.out._op = _Dead("a deferred MinkowskiEngine call recorded before this tensor failed: %s: %s"
                                   % (type(e).__name__, e), e)
        raise


def flush_all():
    for mgr in list(_LIVE):
        flush(mgr)


def pending_ops(mgr):
    return len(mgr._pending)


def _plan_cat_hints(q):
    """zero-copy `me.cat(up, skip)` (res16unet.py:237,247,257,267): when both inputs are results of norms still in the queue,
    the skip's norm allocates the [N, C_up + C_skip] buffer and writes the right-hand columns, the up's norm the left-hand
    ones, and the cat returns the buffer (MinkowskiBatchNorm._cat_slot_for; any condition it cannot honour -> the cat copies).
    Only for skips whose other consumers read row-strided features in place or copy (convolutions)."""
    cats = [op for op in q if op.kind == CAT and len(op.inp) == 2]
    if not cats:
        return
    index = {id(op): i for i, op in enumerate(q)}
    consumers = {}
    for op in q:
        ins = op.inp if op.kind == CAT else ((op.inp,) if op.residual is None else (op.inp, op.residual))
        for t in ins:
            consumers.setdefault(id(t), []).append(op)
    for c in cats:
        up, skip = c.inp
        pa, pb = up._op, skip._op
        if pa is None or pb is None or pa.kind != BN or pb.kind != BN or pa is pb:
            continue
        if index.get(id(pb), 1 << 30) > index.get(id(pa), -1) or pa.cat_into is not None or pb.cat_up or pa.cat_up or pb.cat_into is not None:
            continue
        if len(consumers.get(id(up), ())) != 1:
            continue
        if any(u.kind == BN for u in consumers.get(id(skip), ())):
            continue
        if not (pa.grad and pb.grad) and (pa.grad or pb.grad):
            continue
        pb.cat_up = up._nch()
        pa.cat_into = skip
        STATS["cat_hints"] += 1


def _run(q):
    from .core import cat_now
    from . import block as _block
    _plan_cat_hints(q)
    hooks = UNIT_HOOKS
    ambient = torch.is_grad_enabled()
    n, i = len(q), 0
    while i < n:
        op = q[i]
        if op.grad != ambient:
            with torch.set_grad_enabled(op.grad):
                i += _step(q, i, n, op, hooks, _block, cat_now, False)
        else:
            i += _step(q, i, n, op, hooks, _block, cat_now, True)


def _step(q, i, n, op, hooks, _block, cat_now, may_block):
    kind = op.kind
    if kind == CONV:
        if may_block and op.grad and not hooks:
            took = _block.try_block(q, i, n)
            if took:
                STATS["blocks"] += 1
                return took
        nxt = q[i + 1] if i + 1 < n else None
        bn = nxt.mod if (nxt is not None and nxt.kind == BN and nxt.inp is op.out and nxt.grad == op.grad) else None
        if hooks:
            for pre, _ in hooks:
                pre(op.mod, (op.inp,), {})
        op.mod._forward_now(op.inp, bn=bn, resolved=op.aux, out=op.out)
        if hooks:
            for _, post in hooks:
                post(op.mod, (op.inp,), {}, op.out)
    elif kind == BN:
        if hooks:
            kw = {"relu": op.relu, "residual": op.residual, "cat_up": op.cat_up, "cat_into": op.cat_into}
            for pre, _ in hooks:
                pre(op.mod, (op.inp,), kw)
        op.mod._forward_now(op.inp, op.relu, op.residual, op.cat_up, op.cat_into, out=op.out)
        if hooks:
            for _, post in hooks:
                post(op.mod, (op.inp,), kw, op.out)
    else:
        cat_now(op.inp, out=op.out)
    return 1
