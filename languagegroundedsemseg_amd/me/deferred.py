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
  * a recorded call EXECUTES as soon as no later call can change what it means (`advance`, run whenever a consumer is recorded):
    a norm once its result has a consumer (ReLU / residual flags are then final), a convolution once it is known whether it
    opens a residual block, i.e. the queue holds at most the open tail of one block and the GPU works on a layer while the host
    records the next; reading `.F` (or shape / dtype / ...) of a pending tensor executes whatever is left (`flush`).
    LGS_DEFER_INCREMENTAL=0 keeps everything queued until such a read.
  * the executor looks at what is in front of it: [conv3, norm+relu, conv3, (conv1, norm,) norm+residual(+relu)] runs as the
    whole-block autograd node (me/block.py: lgs_block_forward / lgs_block_backward; an intermediate of such a block that the
    caller kept is recomputed when it is read: class LazyBlock); a norm whose only consumer is the
    `me.cat(up, skip)` recorded right behind it writes straight into the left-hand columns of the concat buffer, the skip half is
    copied in beside it once, and the cat returns the buffer (one copy of the narrow skip half instead of torch.cat's copy of both;
    its backward hands out column slices, read in place by the norms' row-stride support); everything else runs module by
    module with the epilogue flags it collected.

Nothing is re-ordered except the in-place `+=` above, BatchNorm running statistics are updated exactly once per call, module
forward hooks fire at call time as torch defines them (a hook that reads `.F` simply executes what was recorded so far, i.e.
hooks force the call-by-call sequence), `torch.no_grad()` / `enable_grad()` are honoured per recorded call, and switching a
norm between train() and eval() -- or the backend -- executes what is pending first.  A pending result that the caller DROPS
without reading or consuming it executes at that moment (class Op: the result is held weakly and its finaliser drains the queue),
so a discarded `norm(x)` still updates its running statistics when the call is made.  `LGS_DEFER=0` executes every call
immediately (the unfused sequence: same results, tests/test_gpu_reference_calls.py).
"""
import sys
import weakref

import torch

from .. import tuning as _tuning

ENABLED = _tuning.host("DEFER") != 0
INCREMENTAL = _tuning.host("DEFER_INCREMENTAL") != 0

CONV, BN, CAT = 0, 1, 2
_LIVE = weakref.WeakSet()       # managers with a non-empty queue
UNIT_HOOKS = []                 # [(pre(mod, args, kwargs), post(mod, args, kwargs, out))]: test instrumentation of the executor's
#                                 module-by-module path (the teacher-forced parity tests); their presence disables whole-block nodes
STATS = {"flushes": 0, "ops": 0, "blocks": 0, "cat_hints": 0}


class Op:
    """one recorded call.  The result tensor is held WEAKLY (it holds the op): when the caller drops a pending result nobody
    consumed, nothing can read or modify it any more, so the call runs right then (`__call__` is the weak reference's callback)
    -- a dropped `norm(x)` still updates its running statistics when it is made, as it would executed immediately."""
    __slots__ = ("kind", "mod", "inp", "_out", "relu", "residual", "cat_up", "cat_into", "grad", "uses", "aux", "dropped", "mgr", "bs",
                 "__weakref__")

    def __init__(self, kind, mod, inp, out, aux=None):
        self.kind, self.mod, self.inp, self.aux = kind, mod, inp, aux
        self._out = weakref.ref(out, self)
        self.mgr = weakref.ref(out._manager)
        self.relu, self.residual, self.cat_up, self.cat_into = False, None, 0, None
        self.grad = torch.is_grad_enabled()
        self.uses = 0            # recorded consumers of `out`
        self.dropped = False
        self.bs = False          # a convolution that may open a residual block (3^3, stride 1, plain MinkowskiConvolution, grad mode)

    @property
    def out(self):
        return self._out()

    def __call__(self, _ref):
        """the pending result was garbage-collected"""
        self.dropped = True
        mgr = self.mgr()
        if mgr is None or not ENABLED or sys.is_finalizing():
            return
        q = mgr._pending
        if q and any(o is self for o in q):
            try:
                _drain(mgr, q, False)
            except Exception as e:       # (an exception cannot leave a weak-reference callback)
                print("[lgs] a dropped MinkowskiEngine call failed when it was executed: %s: %s" % (type(e).__name__, e), file=sys.stderr)


class _Dead:
    """what `_op` of a tensor becomes when its value will never exist: the queue failed, or the tensor was an intermediate of a
    block that ran as one node"""
    kind = -1

    def __init__(self, why, cause=None):
        self.why, self.cause = why, cause


class LazyBlock:
    """`_op` shared by the intermediates of a residual block that ran as ONE node (me/block.py): conv1's output, norm1's
    (rectified in place), conv2's, and the downsample branch's two.  Their values were never formed -- the reference's own block
    rebinds `out` and never looks at them (resnet_block.py:41-57) -- but a caller that KEPT one (a custom block returning norm1's
    output, a debugger, a feature probe) reads what MinkowskiEngine would have given it: the first read re-runs the block's
    recorded calls module by module from the block's input, up to the tensor asked for, with autograd on (a second branch from the
    same input and weights: its gradient adds to the fused node's, as the chain rule has it) and the norms' running statistics
    put back afterwards (the fused node already counted this batch).  Holds the block input strongly and the intermediates
    weakly, so a block nobody looks into keeps nothing alive once its wrappers are rebound.
    steps: [(kind, module, resolved kernel map | relu flag, index of the input step or -1 for the block input, weakref(out))]"""
    kind = -2
    __slots__ = ("x", "steps")

    def __init__(self, x, steps):
        self.x, self.steps = x, steps

    def materialise(self):
        steps = self.steps
        need = [False] * len(steps)
        for j in range(len(steps) - 1, -1, -1):
            t = steps[j][4]()
            if (t is not None and t._op is self) or need[j]:
                need[j] = True
                if steps[j][3] >= 0:
                    need[steps[j][3]] = True
        kept = []
        for j, (kind, mod, _a, _s, _r) in enumerate(steps):
            if need[j] and kind == BN and mod.bn.track_running_stats:
                b = mod.bn
                kept.append((b, b.running_mean.clone(), b.running_var.clone(), b.num_batches_tracked.clone()))
        vals = {}
        try:
            with torch.enable_grad():
                for j, (kind, mod, arg, src, ref) in enumerate(steps):
                    if not need[j]:
                        continue
                    t = ref()
                    if t is not None and t._op is not self:
                        vals[j] = t                                   # (already has its value)
                        continue
                    inp = self.x if src < 0 else vals[src]
                    vals[j] = mod._forward_now(inp, resolved=arg, out=t) if kind == CONV else mod._forward_now(inp, relu=arg, out=t)
        finally:
            with torch.no_grad():
                for b, rm, rv, nbt in kept:
                    b.running_mean.copy_(rm); b.running_var.copy_(rv); b.num_batches_tracked.copy_(nbt)


_ST = None          # core.SparseTensor / core.cat_now / block: bound on first use (core imports this module)
_CAT_NOW = None
_BLOCK = None


def _bind():
    global _ST, _CAT_NOW, _BLOCK
    from . import core, block
    _ST, _CAT_NOW, _BLOCK = core.SparseTensor, core.cat_now, block


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
    if _ST is None:
        _bind()
    SparseTensor = _ST
    mgr = inp._manager
    meta = inp._meta if inp._op is not None else (None, inp._F.dtype, inp._F.device)
    out = SparseTensor._pending(resolved[0], mgr, (mod.out_channels, meta[1], meta[2]))
    op = out._op = Op(CONV, mod, inp, out, resolved)
    op.bs = op.grad and mod.kernel_volume == 27 and not resolved[2] and mod.stride[0] == 1 and type(mod) is _BLOCK._mod.MinkowskiConvolution
    _use(inp)
    _push(mgr, op)
    if INCREMENTAL:
        advance(mgr)
    return out


def record_bn(mod, inp):
    if _ST is None:
        _bind()
    SparseTensor = _ST
    mgr = inp._manager
    meta = inp._meta if inp._op is not None else (inp._F.shape[1], inp._F.dtype, inp._F.device)
    out = SparseTensor._pending(inp.coordinate_map_key, mgr, meta)
    op = out._op = Op(BN, mod, inp, out)
    _use(inp)
    _push(mgr, op)
    return out


def record_cat(tensors):
    if _ST is None:
        _bind()
    SparseTensor = _ST
    first = tensors[0]
    mgr = first._manager
    ch = sum(t._nch() for t in tensors)
    meta = first._meta if first._op is not None else (None, first._F.dtype, first._F.device)
    out = SparseTensor._pending(first.coordinate_map_key, mgr, (ch, meta[1], meta[2]))
    op = out._op = Op(CAT, None, tuple(tensors), out)
    for t in tensors:
        _use(t)
    _push(mgr, op)
    if INCREMENTAL:
        advance(mgr)
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
    if op.kind == LazyBlock.kind:
        op.materialise()
        if t._op is not None:
            raise RuntimeError("an intermediate of a fused residual block could not be recomputed")
        return
    if op.kind < 0:
        if op.cause is not None:
            raise RuntimeError(op.why) from op.cause
        raise RuntimeError(op.why)
    flush(t._manager)
    if t._op is not None:
        materialise(t)          # dead after the flush: raises


def flush(mgr):
    """execute everything recorded on this manager (a value is needed)"""
    q = mgr._pending
    if not q:
        return
    STATS["flushes"] += 1
    _drain(mgr, q, True)


def advance(mgr):
    """execute the head of the queue as far as no later call can change what it means (INCREMENTAL): a convolution runs once it is
    known not to open a residual block (or the block is complete), a norm once its result has a consumer (its ReLU / residual
    epilogue is then final).  Called whenever a consumer is recorded; the GPU starts on a layer while the host records the next"""
    q = mgr._pending
    n = len(q)
    if n == 0 or (n < 4 and q[0].bs and not q[0].dropped):     # a block needs four calls before anything about it can be decided
        return
    _drain(mgr, q, False)


def _drain(mgr, q, final):
    if mgr._draining:              # re-entered from a weak-reference callback while an operation of this queue is executing
        mgr._redrain = True
        return
    mgr._draining = True
    try:
        while True:
            mgr._redrain = False
            done = _run(q, final)
            STATS["ops"] += done
            if done == len(q):
                del q[:]
                _LIVE.discard(mgr)
                break
            if done:
                del q[:done]
            if not mgr._redrain:
                break
    except BaseException as e:
        failed = list(q)
        del q[:]
        _LIVE.discard(mgr)
        for op in failed:
            t = op.out
            if t is not None and t._op is not None and t._op.kind >= 0:
                t._op = _Dead("a deferred MinkowskiEngine call recorded before this tensor failed: %s: %s"
                              % (type(e).__name__, e), e)
        raise
    finally:
        mgr._draining = False


def flush_all():
    for mgr in list(_LIVE):
        flush(mgr)


def pending_ops(mgr):
    return len(mgr._pending)


def _cat_partner(q, i, n, op):
    """the `me.cat(up, skip)` (res16unet.py:237,247,257,267) this norm's result goes to as the FIRST input, if that is its only
    consumer and the skip tensor exists already: the norm then writes straight into the left-hand columns of the [N, C_up +
    C_skip] concat buffer (lgs_bn_forward's output row stride), the skip half is copied in once, and the cat returns the buffer
    -- the concat costs one copy of the skip half instead of a copy of both halves (MinkowskiBatchNorm._cat_slot_for; any
    condition it cannot honour -> the cat copies both)."""
    if op.uses != 1 or op.cat_into is not None:
        return None
    for j in range(i + 1, min(i + 3, n)):
        c = q[j]
        if c.kind == CAT and len(c.inp) == 2 and c.inp[0] is op.out:
            skip = c.inp[1]
            if skip._op is None and c.grad == op.grad:
                return c
            return None
    return None


def _run(q, final):
    """-> number of queue entries executed (all of them when `final`)"""
    if _ST is None:
        _bind()
    cat_now, _block = _CAT_NOW, _BLOCK
    hooks = UNIT_HOOKS
    ambient = torch.is_grad_enabled()
    n, i = len(q), 0
    while i < n:
        op = q[i]
        if op.grad != ambient:
            with torch.set_grad_enabled(op.grad):
                took = _step(q, i, n, op, hooks, _block, cat_now, final)
        else:
            took = _step(q, i, n, op, hooks, _block, cat_now, final)
        if took <= 0:
            break
        i += took
    return i


def _step(q, i, n, op, hooks, _block, cat_now, final):
    """execute q[i] (or the block it opens) -> entries consumed; 0 = not yet (only when not `final`)"""
    kind = op.kind
    if kind == CONV:
        if not hooks:
            took = _block.try_block(q, i, n, final)
            if took > 0:
                STATS["blocks"] += 1
                return took
            if took < 0:
                return 0
        nxt = q[i + 1] if i + 1 < n else None
        bn = nxt.mod if (nxt is not None and nxt.kind == BN and nxt.inp is op.out and nxt.grad == op.grad) else None
        if hooks:
            for pre, _ in hooks:
                pre(op.mod, (op.inp,), {})
        out = op.out
        res = op.mod._forward_now(op.inp, bn=bn, resolved=op.aux, out=out)
        if hooks:
            for _, post in hooks:
                post(op.mod, (op.inp,), {}, res)
    elif kind == BN:
        if op.uses == 0 and not final and not op.dropped:
            return 0                       # still open: an in-place ReLU or `+=` may follow
        cat = _cat_partner(q, i, n, op)
        if cat is not None:
            op.cat_into = cat.inp[1]
            STATS["cat_hints"] += 1
        if hooks:
            kw = {"relu": op.relu, "residual": op.residual, "cat_up": op.cat_up, "cat_into": op.cat_into}
            for pre, _ in hooks:
                pre(op.mod, (op.inp,), kw)
        res = op.mod._forward_now(op.inp, op.relu, op.residual, op.cat_up, op.cat_into, out=op.out)
        if hooks:
            for _, post in hooks:
                post(op.mod, (op.inp,), kw, res)
    else:
        out = op.out
        if out is not None:                # (a dropped concat has no side effect: nothing to do)
            cat_now(op.inp, out=out)
    return 1
