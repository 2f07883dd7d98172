"""Raw op wrappers: torch tensors in, C-ABI calls on torch's current HIP stream out.

Plumbing only (pointers, strides, shape checks).  No arithmetic happens here and nothing falls back to
ATen: if the library or a GPU is missing these raise."""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID  # noqa: F401

LEAK = 0.01
NORM_EPS = 1e-5
MODE_IN, MODE_BN_TRAIN, MODE_BN_EVAL, MODE_GN = 0, 1, 2, 3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    if t.dtype == torch.float32:
        return L.XH_F32
    if t.dtype == torch.bfloat16:
        return L.XH_BF16
    if t.dtype == torch.float16:
        return L.XH_F16
    raise TypeError(f"activation dtype must be float32, bfloat16 or float16, got {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


def _vol(t):
    """(N, C, D, H, W, batch_stride) of a tensor whose samples are contiguous CDHW blocks."""
    if t.dim() != 5:
        raise ValueError(f"expected a 5D NCDHW tensor, got {tuple(t.shape)}")
    if not t.is_cuda:
        raise RuntimeError("xlstm_hved_amd ops need device tensors (the HIP library is the only compute path)")
    n, c, d, h, w = t.shape
    if not t[0].is_contiguous():
        raise ValueError("each sample must be a contiguous (C,D,H,W) block")
    bs = t.stride(0) if n > 1 else c * d * h * w
    return n, c, d, h, w, bs


def _f32(t, what):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{what} must be a contiguous float32 device tensor")
    return t


def _arr4(tensors):
    a = (C.c_void_p * L.MAX_WPTR)()
    for i, t in enumerate(tensors):
        a[i] = _p(t)
    return a


_MFMA = [True]


def set_mfma(enabled):
    """A/B switch for the bf16-MFMA conv kernels (default on)."""
    _MFMA[0] = bool(enabled)
    L.check(L.load().xh_set_option(0, int(bool(enabled))), "xh_set_option")


# Arithmetic of fp32 STORAGE (xh_conv_desc.arith): part of every conv call, not library state.
#   _FP32_MFMA[0]  the default of this process' Python side (set_fp32_mfma), used by calls made outside any scope;
#   ARITH[0]       the mode of the calls being issued right now: None = the default, else an arith bit mask.  A model with a
#                  `fp32_arith` attribute ("split" / "vector") runs its forward inside arith_scope(...), and every
#                  functional.Function remembers the mode of its forward for its backward -- so two models of different modes can
#                  interleave forwards and backwards in one process, and a captured graph keeps the mode it was captured with.
_FP32_MFMA = [False]
ARITH = [None]
ARITH_NAMES = {None: None, "vector": 0, "split": L.ARITH_F32_SPLIT, "split_k7vector": L.ARITH_F32_SPLIT | L.ARITH_K7_VECTOR}


def set_fp32_mfma(enabled):
    """Default arithmetic of fp32 STORAGE for calls outside an arith_scope: the quad-channel matrix-core kernels with two-term fp16
    operands (xh_conv_desc.arith = XH_ARITH_F32_SPLIT: csrc/conv3d_q4s.hip for k = 3 stride-1 convs and their data gradients, the
    16-bit-operand weight-gradient kernels, the 7^3 gate convs with operands rounded once to fp16) instead of the fp32 vector
    kernels; default off.  ~22-bit products in the forward pass; in the backward pass the activation gradients must sit in
    fp16's range: scale the loss as for fp16 storage (bench.py and TrainStep use 65536).  Python-side state only: the library
    takes the mode with every call."""
    _FP32_MFMA[0] = bool(enabled)


def current_arith():
    """The xh_conv_desc.arith value of a conv call issued now."""
    a = ARITH[0]
    return (L.ARITH_F32_SPLIT if _FP32_MFMA[0] else 0) if a is None else a


class arith_scope:
    """with arith_scope("split" | "vector" | "split_k7vector" | None | <int>): conv calls inside take that fp32-storage arithmetic
    (None: leave whatever is current)."""

    def __init__(self, mode):
        self.mode = ARITH_NAMES[mode] if (mode is None or isinstance(mode, str)) else int(mode)

    def __enter__(self):
        self.prev = ARITH[0]
        if self.mode is not None:
            ARITH[0] = self.mode
        return self

    def __exit__(self, *exc):
        ARITH[0] = self.prev
        return False


MIXED = [None]


def set_mixed_storage(dtype):
    """Mixed storage policy for fp32 inputs: None (default: every tensor in the input's type), or torch.float16 / torch.bfloat16 --
    the ENCODER half (init blocks, encoders, skip-return path, DRBs) keeps fp32 storage, the decoder half (PoE onwards) stores in
    this 16-bit type.  tools/precision_sweep.py: the mask flips of 16-bit storage come from the encoder trunk (every rounding
    there is amplified by all the InstanceNorms behind it), not from the decoders."""
    if dtype not in (None, torch.float16, torch.bfloat16):
        raise ValueError("mixed storage: None, torch.float16 or torch.bfloat16")
    MIXED[0] = dtype


def last_conv_kernel():
    """Template instance launched by the most recent conv3d / conv3d_wgrad call (bench.py attributes timings with it)."""
    return L.load().xh_last_conv_kernel().decode()


def new_like(t, shape, dtype=None):
    return torch.empty(shape, dtype=dtype or t.dtype, device=t.device)


# fp64 reduction scratch.  Every fused reduction (norm moments, norm-backward sums, gate sums) accumulates into a
# zeroed (n, c, 2) fp64 buffer that is consumed a launch or two later inside the same autograd Function.  A step needs
# >100 of them; handing them out of one arena that the model zeroes with ONE fill at the start of forward() replaces >100
# tiny fill launches per step.  Without a reset (stages called on their own) the arena runs out and fresh zeros are used.
_ARENA_DOUBLES = 1 << 19          # 4 MiB
_ARENA = {}


def zeros_f64(device, shape):
    numel = 1
    for s_ in shape:
        numel *= int(s_)
    a = _ARENA.get(device)
    if a is None:
        a = _arenas(device)[0]
    need = (numel + 15) & ~15
    if a[1] + need > _ARENA_DOUBLES:
        return torch.zeros(shape, dtype=torch.float64, device=device)
    t = a[0][a[1]:a[1] + numel].view(shape)
    a[1] += need
    a[2] = max(a[2], a[1])
    return t


# fp32 twin of the arena: zero-initialised gradient buffers of weights that are not leaf parameters (composed 7^3 / head
# weights: their gradients go through autograd to the compose kernels) -- ~11 small fills per backward otherwise.  Slices
# are handed out during backward and consumed before it ends; the reset at the start of the next forward zeroes them again.
_ARENA32_FLOATS = 1 << 20         # 4 MiB
_ARENA32 = {}
_ARENA_BYTES = {}                 # both arenas of a device are one allocation (fp64 part first): ONE fill zeroes them


def _arenas(device):
    raw = _ARENA_BYTES[device] = torch.zeros(_ARENA_DOUBLES * 8 + _ARENA32_FLOATS * 4, dtype=torch.uint8, device=device)
    _ARENA[device] = [raw[:_ARENA_DOUBLES * 8].view(torch.float64), 0, 0]
    _ARENA32[device] = [raw[_ARENA_DOUBLES * 8:].view(torch.float32), 0, 0]
    return _ARENA[device], _ARENA32[device]


def zeros_f32(device, numel):
    a = _ARENA32.get(device)
    if a is None:
        a = _arenas(device)[1]
    need = (int(numel) + 15) & ~15
    if a[1] + need > _ARENA32_FLOATS:
        return torch.zeros(int(numel), dtype=torch.float32, device=device)
    t = a[0][a[1]:a[1] + int(numel)]
    a[1] += need
    a[2] = max(a[2], a[1])
    return t


def red_arena_reset(device):
    """Zeroes the arena up to its HIGH-WATER mark (one fill) and starts over.  Called where no arena slice is live: the
    start of the network's forward().  The high-water mark (largest extent ever handed out, captures included) rather than
    the current offset: a hipGraph replay dirties the extent it had at capture time without moving the Python-side
    offset, so a smaller eager pass in between must not shrink what the next reset clears."""
    # Under stream capture the WHOLE arena is zeroed: the captured fill must cover every slice the captured step will
    # dirty, and the high-water mark only knows what ran before (a step captured without a preceding eager pass of the same
    # shape would otherwise leave stale sums / gradients for replays 2..N).
    whole = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    a, b, raw = _ARENA.get(device), _ARENA32.get(device), _ARENA_BYTES.get(device)
    if raw is None:
        return
    if whole:
        raw.zero_()                                    # one fill for both arenas
    elif b[2] > 0:
        raw[:_ARENA_DOUBLES * 8 + b[2] * 4].zero_()    # the fp64 arena and the used part of the fp32 one, still one fill
    elif a[2] > 0:
        a[0][:a[2]].zero_()
    a[1] = 0
    b[1] = 0


def zeros_red(t, n, c):
    return zeros_f64(t.device, (n, c, 2))


# Statistics fan-in workspaces (xh_conv_ptrs.fan, csrc/fanin.h): the library owns no device memory, so the zero-initialised
# block a statistics-producing launch sums through is handed in here -- one per (device, stream): launches ordered on a
# stream never overlap and share it, launches on different streams get different blocks, every capture gets its own.
_FAN = {}


def begin_capture_scope():
    """Call right before a stream capture starts: the capture gets a statistics fan-in block of its own (fan_block).  Without it
    a capture is recognised by the first launch that finds the stream capturing after an eager launch."""
    for per in _FAN.values():
        per["in_capture"] = False


def fan_block(device):
    """(pointer, bytes) of the zero-initialised fan-in block for a launch on the current stream (csrc/fanin.h: one block per
    chain of launches that can be in flight at the same time, every launch leaves it zero).

    Eager launches: one block per (device, stream).  Captured launches: one block per CAPTURE -- two captured graphs may be
    replayed on different streams at once (bench's forward graph next to TrainStep's, a replay overlapped with a communication
    stream).  A capture cannot allocate-and-zero without putting a fill into the graph, so blocks are taken from a small stock
    of pre-zeroed spares that eager launches keep topped up; they are never freed (graphs hold their addresses).  No spare
    left: (None, 0) -- the launch keeps its direct atomics, which is slower but exact."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    per = _FAN.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if per is None:
        if capturing:
            return None, 0
        nbytes = int(L.load().xh_fanin_bytes())
        per = _FAN[key] = {"bytes": nbytes, "streams": {}, "spares": [], "used": [], "cap": None, "in_capture": False}
    if capturing:
        sid = torch.cuda.current_stream(device).cuda_stream
        if not per["in_capture"] or per["cap"] is None:
            per["in_capture"] = True
            per["cap"] = per["spares"].pop() if per["spares"] else None
            per["cap_stream"] = sid
            if per["cap"] is not None:
                per["used"].append(per["cap"])
        # The capture's block belongs to the stream the capture started on: launches captured on a side stream become parallel
        # branches of the graph (set_level_streams) and could run next to the origin stream's statistics launches -- the block's
        # contract is ONE launch in flight -- so they keep their direct atomics.
        if per["cap"] is None or sid != per.get("cap_stream"):
            return None, 0
        return per["cap"].data_ptr(), per["bytes"]
    per["in_capture"] = False
    while len(per["spares"]) < 4:
        per["spares"].append(torch.zeros(per["bytes"], dtype=torch.uint8, device=device))
    sid = torch.cuda.current_stream(device).cuda_stream
    blk = per["streams"].get(sid)
    if blk is None:
        blk = per["streams"][sid] = torch.zeros(per["bytes"], dtype=torch.uint8, device=device)
    return blk.data_ptr(), per["bytes"]


# ------------------------------------------------------------------------------------- weight fragments
# The MFMA conv kernels read their weights as packed 16-bit fragments (xh_conv3d_workspace_bytes).  Weights are constant
# within a step, so instead of one small pack launch in front of each of the ~54 k=3 convolutions (forward and data
# gradient) the network's forward() calls prepack_all(): ONE launch per 24 convolutions packs every registered conv's
# fragments into its persistent workspace.  A conv call registers itself the first time it is seen (and packs on its own
# that time).  An entry is trusted only while (a) it was packed in the current epoch (prepack_all starts a new one) and
# (b) the version counters of its weight tensors are the ones it was packed at -- anything else packs in front of the conv
# exactly as before, so a stage called on its own or a weight changed mid-step can never read stale fragments.
_PACKS = {}
_PACK_STATE = {"epoch": 0, "arrays": None, "enabled": True}


class _PackEntry:
    __slots__ = ("refs", "ws", "desc", "ptrs", "epoch", "versions", "keep", "dptrs")

    def alive(self):
        """The weight tensors still exist AND still own the storage whose addresses the entry recorded (model.to() /
        .half() / `p.data = ...` replace the storage under a surviving Parameter: the raw pointers would dangle)."""
        for r, dp in zip(self.refs, self.dptrs):
            t = r()
            if t is None or t.data_ptr() != dp:
                return False
        return True


def _wversion(t):
    """What tells a stale packed copy of a weight tensor from a fresh one: autograd's version counter for parameters; for the
    tensors functional.ComposeAll writes with a raw kernel into persistent storage (no version bump) the generation it stamps on
    them (`_xh_gen`, on the alias and on its storage)."""
    g = getattr(t, "_xh_gen", None)
    return t._version if g is None else ("gen", g)


_PACK_TABLE = [os.environ.get("XH_NO_PACK_TABLE", "") == ""]     # A/B switch: one launch through a device-resident job table


def set_pack_table(enabled):
    _PACK_TABLE[0] = bool(enabled)
    _PACK_STATE["table"] = None


def set_prepack(enabled):
    """A/B switch (tests, microbenchmarks): False = every conv packs its own fragments right in front of the launch."""
    _PACK_STATE["enabled"] = bool(enabled)
    _PACKS.clear()
    _PACK_STATE["arrays"] = None


def _pack_entry(weights, desc, need, device):
    key = (tuple(w.data_ptr() for w in weights), desc.dtype, desc.arith, desc.Cin, desc.Cout, desc.groups, desc.transposed, desc.D,
           desc.H, desc.W, desc.n_wptr, need)
    e = _PACKS.get(key)
    if e is not None and e.alive():
        return e
    import weakref
    e = _PackEntry()
    # (a tensor composed by functional.ComposeAll is a per-step alias of persistent storage: the entry follows the storage)
    e.refs = [weakref.ref(getattr(w, "_xh_base", w)) for w in weights]
    e.dptrs = [w.data_ptr() for w in weights]
    e.ws = torch.empty(need, dtype=torch.uint8, device=device)
    e.desc = L.ConvDesc.from_buffer_copy(desc)
    e.ptrs = L.ConvPtrs()
    e.ptrs.w = _arr4(list(weights))
    e.ptrs.ws, e.ptrs.ws_bytes = e.ws.data_ptr(), need
    e.epoch, e.versions = -1, None
    _PACKS[key] = e
    _PACK_STATE["arrays"] = None
    return e


def _pack_arrays_refresh():
    st = _PACK_STATE
    if st["arrays"] is None or any(not e.alive() for e in st["arrays"][0]):
        for k in [k for k, e in _PACKS.items() if not e.alive()]:
            del _PACKS[k]
        ents = list(_PACKS.values())
        n = len(ents)
        st["arrays"] = (ents, (C.c_void_p * n)(*[C.addressof(e.desc) for e in ents]),
                        (C.c_void_p * n)(*[C.addressof(e.ptrs) for e in ents]))
        st["table"] = None


def _pack_table_refresh():
    """ONE launch through a device-resident job table (xh_conv3d_prepack_table): built on the host when the set of convolutions
    changes and copied to the device OUTSIDE any capture; a set that changes under capture takes the kernel-argument launches."""
    st = _PACK_STATE
    ents, darr, parr = st["arrays"]
    if st.get("table") is None and ents and not torch.cuda.is_current_stream_capturing() and _PACK_TABLE[0]:
        lib = L.load()
        nbytes = int(lib.xh_conv3d_prepack_table_bytes())
        host = (C.c_char * nbytes)()
        if lib.xh_conv3d_prepack_table(len(ents), darr, parr, C.cast(host, C.c_void_p)) == 0:
            head = (C.c_int * 2).from_buffer(host)
            dev = torch.frombuffer(host, dtype=torch.uint8).to(ents[0].ws.device)
            st["table"] = (dev, int(head[1]), ents[0].ws.device)
            # never freed: a captured graph replays the table it was captured with (like the fan-in blocks); 40 KB each
            st.setdefault("tables_keep", []).append(dev)


def prepare_capture():
    """Call right BEFORE a stream capture of a step that has run eagerly at least once: what a captured launch cannot do for itself
    is done here -- the weight-pack job table of every convolution registered so far is built and copied to the device (inside
    the capture prepack_all would fall back to its three kernel-argument launches), and the capture gets a statistics fan-in
    block of its own (begin_capture_scope)."""
    begin_capture_scope()
    if _PACK_STATE["enabled"] and _PACKS and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        _pack_arrays_refresh()
        if _PACK_STATE["arrays"][0]:
            _pack_table_refresh()


def prepack_all():
    """Starts a new epoch and packs the fragments of every registered convolution (xh_conv3d_prepack).  Called at the start
    of the network's forward(); capture-safe (the launches are part of a captured step)."""
    st = _PACK_STATE
    st["epoch"] += 1
    if not st["enabled"] or not _PACKS:
        return
    _pack_arrays_refresh()
    ents, darr, parr = st["arrays"]
    if not ents:
        return
    lib = L.load()
    _pack_table_refresh()
    tab = st.get("table")
    if tab is not None and tab[2] == ents[0].ws.device:
        L.check(lib.xh_conv3d_prepack_run(_stream(), tab[0].data_ptr(), tab[1]), "xh_conv3d_prepack_run")
    else:
        L.check(lib.xh_conv3d_prepack(_stream(), len(ents), darr, parr), "xh_conv3d_prepack")
    for e in ents:
        e.epoch = st["epoch"]
        e.versions = tuple(_wversion(r()) for r in e.refs)


# ----------------------------------------------------------------------------------------------- conv
def _conv_desc(xa, xb, k, stride, groups, cout, n_w, transposed, pre, act, act_slope, epi, e, y_bs, out_sp):
    n, ca, d, h, w, xa_bs = _vol(xa)
    cb, xb_bs = 0, 0
    if xb is not None:
        n2, cb, d2, h2, w2, xb_bs = _vol(xb)
        if (n2, d2, h2, w2) != (n, d, h, w) or xb.dtype != xa.dtype:
            raise ValueError("second conv source does not match the first")
    desc = L.ConvDesc()
    desc.dtype = _dt(xa)
    desc.arith = current_arith()
    desc.N, desc.Cin, desc.Cout, desc.groups = n, ca + cb, cout, groups
    desc.D, desc.H, desc.W = d, h, w
    desc.Do, desc.Ho, desc.Wo = out_sp
    desc.k, desc.stride, desc.Ca = k, stride, ca
    desc.xa_bs, desc.xb_bs, desc.y_bs = xa_bs, xb_bs, y_bs
    desc.n_wptr, desc.transposed = n_w, int(transposed)
    desc.pre = int(pre is not None)
    desc.pre_slope = float(pre[2]) if pre is not None else 1.0
    desc.act, desc.act_slope, desc.epi = act, act_slope, epi
    if e is not None:
        ea, eb = e[0], e[1]
        desc.Cea = ea.shape[1]
        desc.ea_bs = _vol(ea)[5]
        desc.eb_bs = _vol(eb)[5] if eb is not None else 0
        desc.e_slope = float(e[4])
    return desc


def _out_spatial(d, h, w, k, stride):
    pad = k // 2
    f = lambda s: (s + 2 * pad - k) // stride + 1
    return f(d), f(h), f(w)


def _check_weights(weights, biases, cin, cout, groups, k, transposed, what):
    """The kernels index weights from (Cin, Cout, groups, k) alone: a tensor of another shape would be read out of bounds."""
    nw = len(weights)
    if nw != 1 and nw != groups:
        raise ValueError(f"{what}: {nw} weight tensors for {groups} groups (need 1 or one per group)")
    if cin % groups or cout % groups:
        raise ValueError(f"{what}: channels ({cin} -> {cout}) not divisible by groups={groups}")
    rows, cols = (cin, cout) if transposed else (cout, cin)      # transposed: forward-layout weights [Cin_fwd... ] = [rows][cols/groups]
    want = (rows // nw, cols // groups, k, k, k)
    for w in weights:
        if tuple(w.shape) != want:
            raise ValueError(f"{what}: weight shape {tuple(w.shape)}, expected {want} (Cin={cin}, Cout={cout}, groups={groups}, k={k})")
    for b in biases or []:
        if b is not None and tuple(b.shape) != (cout // nw,):
            raise ValueError(f"{what}: bias shape {tuple(b.shape)}, expected {(cout // nw,)}")


PAIR = [os.environ.get("XH_NO_PAIR", "") == ""]             # A/B switch: the decoder's recon | seg pair launches (model._forward_pair)


LEVEL_STREAMS = [False]


def set_level_streams(enabled):
    """A/B switch, eager launches only: the three coarse latent-path chains (PoE output -> VU block -> upsampling -> conv block,
    RA_HVED.py:599-603) on side streams next to the finest one (they are independent of each other until the decoders).  Inert
    under stream capture (model._decode says why)."""
    LEVEL_STREAMS[0] = bool(enabled)


def set_pair(enabled):
    PAIR[0] = bool(enabled)


SEP_COMPOSE = [os.environ.get("XH_NO_SEP_COMPOSE", "") == ""]      # A/B switch: depthwise o pointwise of the skip-return ResBlocks as one dense conv


def set_sep_compose(enabled):
    SEP_COMPOSE[0] = bool(enabled)


STREAM5 = [os.environ.get("XH_NO_STREAM5", "") == ""]     # A/B switch: the skip stream as a fifth group of the encoder launches (model._encode5)


def set_stream5(enabled):
    STREAM5[0] = bool(enabled)


NB_PENDING = {}        # functional.InLreluConv: gradients handed over unwritten, by address (see functional._NB_PENDING)
def conv3d_fuses_bn(x, cout, k=3, groups=1):
    """True when conv3d(x, ..., in_stats=(red, count, slope, bn)) can finalise a training-mode BatchNorm of x inside its launch."""
    if not (_BN_FOLD_ON() and _MFMA[0] and x.is_cuda and x.shape[0] == 1):
        return False
    d = _conv_desc(x, None, k, 1, groups, cout, 1, False, (None, None, 0.0), ACT_NONE, LEAK, 0, None, 0, _out_spatial(*x.shape[2:], k, 1))
    return bool(L.load().xh_conv3d_fuses_bn_finalize(C.byref(d)))


def _BN_FOLD_ON():
    return BN_FOLD[0]


_NB_FOLD = [os.environ.get("XH_NO_NB_FOLD", "") == ""]      # A/B switch: the InstanceNorm backward folded into the consuming data gradient


def set_norm_bwd_fold(enabled):
    _NB_FOLD[0] = bool(enabled)


def conv3d(xa, xb, weights, biases, *, k, cout, stride=1, groups=1, transposed=False, pre=None, act=ACT_NONE,
           act_slope=LEAK, epi=0, e=None, red=None, out=None, in_stats=None, nb=None, bcast=0):
    """y = act(conv(pre(cat[xa, xb])) + b) [+ fused epilogue].  weights/biases: lists of 1 or `groups` fp32 tensors.
    pre = (sc, sh, slope); e = (ea, eb, e_sc, e_sh, e_slope) for epi==1.  `out` may be a channel slice.
    in_stats = (red, count, slope) instead of pre: InstanceNorm + LeakyReLU of the input from its raw channel sums; returns
    (y, sc, sh, mean, rstd).  On the MFMA path the finalisation rides on the weight-pack launch, else xh_norm_finalize runs.
    nb = (px, nb_red, mean, rstd, pd): xa is the masked data gradient g of the stage behind this conv and the conv's real input
    is that stage's InstanceNorm backward, A*g + C*px + B (xh_conv_desc.pre == 2): applied on load where the kernel can, and
    stored into `pd` (a tensor of xa's shape) either way -- by this launch, or by an xh_in_bwd_apply pass in front of it.
    bcast = 4 (xh_conv_desc.bcast): a broadcast operand -- forward: xa has cin / 4 channels, each read by the four input channels of a
    group through their own pre (sc, sh); transposed with epi == 1: e[0] has cout / 4 channels, and out=False stores nothing (the
    masked gradient is only summed into `red`; returns None)."""
    lib = L.load()
    if bcast and (xb is not None or nb is not None or in_stats is not None or bcast != 4 or (out is False and not (transposed and epi == 1))):
        raise ValueError("conv3d: a broadcast operand takes one source, pre = (sc, sh, slope), bcast = 4")
    if nb is not None:
        px, nb_red, nb_mean, nb_rstd, pd = nb
        if xb is not None or pre is not None or in_stats is not None:
            raise ValueError("conv3d: nb excludes xb / pre / in_stats")
        dsc = _conv_desc(xa, None, k, stride, groups, cout, len(weights), transposed, None, act, act_slope, epi, e, 0,
                         _out_spatial(*xa.shape[2:], k, stride))
        dsc.pre, dsc.px_bs, dsc.pd_bs = 2, _vol(px)[5], _vol(pd)[5]
        if not (_NB_FOLD[0] and _MFMA[0] and lib.xh_conv3d_fuses_norm_bwd(C.byref(dsc))):
            in_bwd_apply(xa, px, nb_red, nb_mean, nb_rstd, have_g=True, out=pd)
            xa, nb = pd, None
    n, ca, d, h, w, _ = _vol(xa)
    cin_l = ca * bcast if (bcast and not transposed) else ca + (xb.shape[1] if xb is not None else 0)     # logical input channels
    _check_weights(weights, biases, cin_l, cout, groups, k, transposed, "conv3d")
    osp = _out_spatial(d, h, w, k, stride)
    if out is None:
        out = new_like(xa, (n, cout) + osp)
    y_bs = _vol(out)[5] if out is not False else 0
    if out is not False and (tuple(out.shape) != (n, cout) + osp or out.dtype != xa.dtype):
        raise ValueError(f"conv output tensor has shape {tuple(out.shape)}, expected {(n, cout) + osp}")
    stats = None
    if in_stats is not None:
        cin = ca + (xb.shape[1] if xb is not None else 0)
        stats = tuple(torch.empty((n, cin), dtype=torch.float32, device=xa.device) for _ in range(4))   # sc, sh, mean, rstd
        pre = (stats[0], stats[1], in_stats[2])
    desc = _conv_desc(xa, xb, k, stride, groups, cout, len(weights), transposed, pre, act, act_slope, epi, e, y_bs, osp)
    if bcast:
        desc.bcast = bcast
        if transposed:
            if e is None or e[0].shape[1] * bcast != cout or e[1] is not None or e[0].dtype != xa.dtype:
                raise ValueError("conv3d: the broadcast e operand has cout / 4 channels and one source")
            desc.Cea = cout
        else:
            desc.Cin = desc.Ca = cin_l
        if pre is not None and (pre[0].numel() != n * cin_l or pre[1].numel() != n * cin_l):
            raise ValueError("conv3d: pre of a broadcast input is per LOGICAL channel")
    ptrs = L.ConvPtrs()
    ptrs.xa, ptrs.xb = _p(xa), _p(xb)
    if nb is not None:
        desc.pre, desc.px_bs, desc.pd_bs = 2, _vol(px)[5], _vol(pd)[5]
        ptrs.px, ptrs.pd = _p(px), _p(pd)
        ptrs.nb_red, ptrs.nb_mean, ptrs.nb_rstd = _p(nb_red), _p(_f32(nb_mean, "nb_mean")), _p(_f32(nb_rstd, "nb_rstd"))
        ptrs.nb_count = d * h * w
    ptrs.w = _arr4([_f32(t, "conv weight") for t in weights])
    ptrs.b = _arr4([_f32(t, "conv bias") for t in (biases or [])])
    if pre is not None:
        ptrs.pre_sc, ptrs.pre_sh = _p(_f32(pre[0], "pre_sc")), _p(_f32(pre[1], "pre_sh"))
    ptrs.y = _p(out) if out is not False else None
    if e is not None:
        ptrs.ea, ptrs.eb = _p(e[0]), _p(e[1])
        ptrs.e_sc, ptrs.e_sh = _p(_f32(e[2], "e_sc")), _p(_f32(e[3], "e_sh"))
        if len(e) > 5 and e[5] is not None:                 # (broadcast e operand: the centre of the second sum)
            ptrs.e_ctr = _p(_f32(e[5], "e_ctr"))
    if red is not None:
        ptrs.red = _p(red)
        if epi:
            ptrs.fan, ptrs.fan_bytes = fan_block(xa.device)
    need = lib.xh_conv3d_workspace_bytes(C.byref(desc)) if _MFMA[0] else 0
    if need and _PACK_STATE["enabled"]:
        ent = _pack_entry(weights, desc, need, xa.device)
        ptrs.ws, ptrs.ws_bytes = ent.ws.data_ptr(), need
        vers = tuple(_wversion(w) for w in weights)
        if ent.epoch == _PACK_STATE["epoch"] and ent.versions == vers:
            ptrs.ws_packed = 1
        else:                                   # this call packs; good for the rest of the epoch
            ent.epoch, ent.versions = _PACK_STATE["epoch"], vers
    elif need:
        ws = torch.empty(need, dtype=torch.uint8, device=xa.device)
        ptrs.ws, ptrs.ws_bytes = ws.data_ptr(), need
    if stats is not None:
        bn = in_stats[3] if len(in_stats) > 3 else None       # (gamma, beta, running_mean, running_var, steps): BatchNorm flavour
        if bn is not None:
            if not (n == 1 and lib.xh_conv3d_fuses_bn_finalize(C.byref(desc))):
                raise ValueError("conv3d: the BatchNorm flavour of the fused finalisation needs one sample on the quad-channel kernel "
                                 "(ops.conv3d_fuses_bn)")
            ptrs.fin_gamma, ptrs.fin_beta, ptrs.fin_rm, ptrs.fin_rv, ptrs.fin_steps = _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), int(bn[4])
        if need or (k == 3 and stride == 2):   # MFMA path / stride-2 convs: the conv launch finalises the statistics itself
            ptrs.fin_red, ptrs.fin_mean, ptrs.fin_rstd, ptrs.fin_count = _p(in_stats[0]), _p(stats[2]), _p(stats[3]), int(in_stats[1])
        else:
            L.check(lib.xh_norm_finalize(_stream(), 0, _p(in_stats[0]), n, stats[0].shape[1], int(in_stats[1]), 1, NORM_EPS, None,
                                         None, None, None, 1, _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3])),
                    "xh_norm_finalize")
    if _C1_COLLECT[0] is not None and k == 1 and stride == 1 and stats is None and nb is None:
        _C1_COLLECT[0].append((desc, ptrs, (xa, xb, out, weights, biases, pre, e, red)))     # launched by conv1x1_flush()
        return out
    if _PAIR_COLLECT[0] is not None and k == 3 and stride == 1 and not bcast and nb is None and out is not False:
        # launched by conv_pair_scope.__exit__: two independent convs of one shape share a launch (xh_conv3d_fwd_pair)
        _PAIR_COLLECT[0].append((desc, ptrs, (xa, xb, out, weights, biases, pre, e, red, stats)))
    else:
        L.check(lib.xh_conv3d_fwd(_stream(), C.byref(desc), C.byref(ptrs)), "xh_conv3d_fwd")
    if stats is not None:
        return (out,) + stats
    return out if out is not False else None


_PAIR_COLLECT = [None]
CONV_PAIRS = [os.environ.get("XH_NO_CONV_PAIRS", "") == ""]     # A/B switch


def set_conv_pairs(enabled):
    CONV_PAIRS[0] = bool(enabled)


class conv_pair_scope:
    """with conv_pair_scope(): the k = 3 stride-1 conv3d calls inside are recorded (their output tensors are returned as usual) and
    issued when the scope ends -- two of them as ONE launch when they are independent convolutions of one shape
    (xh_conv3d_fwd_pair: the recon | seg streams' first decoder convs), else one by one in the order they were made.  The caller
    guarantees that nothing inside the scope reads an output of a recorded call."""

    def __enter__(self):
        self.on = CONV_PAIRS[0] and _PAIR_COLLECT[0] is None
        if self.on:
            _PAIR_COLLECT[0] = []
        return self

    def __exit__(self, et, ev, tb):
        if not self.on:
            return False
        calls, _PAIR_COLLECT[0] = _PAIR_COLLECT[0], None
        if et is not None:
            return False                                  # (an error inside the scope: nothing is launched)
        lib = L.load()
        i = 0
        while i < len(calls):
            if i + 1 < len(calls):
                rc = lib.xh_conv3d_fwd_pair(_stream(), C.byref(calls[i][0]), C.byref(calls[i][1]), C.byref(calls[i + 1][0]), C.byref(calls[i + 1][1]))
                if rc == 0:
                    i += 2
                    continue
                if rc < 0:
                    L.check(rc, "xh_conv3d_fwd_pair")
            L.check(lib.xh_conv3d_fwd(_stream(), C.byref(calls[i][0]), C.byref(calls[i][1])), "xh_conv3d_fwd")
            i += 1
        return False


def conv3d_supports_bcast(x, cout, groups):
    """True when the broadcast form of the k = 3 conv on x (x: one stored channel per group, 4 logical channels each; cout = 4 *
    groups) is served in all three directions (xh_conv3d_supports_bcast)."""
    if not (_MFMA[0] and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == groups and cout == 4 * groups):
        return False
    n, ca, d, h, w, bs = _vol(x)
    desc = L.ConvDesc()
    desc.dtype, desc.arith, desc.bcast = _dt(x), current_arith(), 4
    desc.N, desc.Cin, desc.Cout, desc.groups = n, 4 * ca, cout, groups
    desc.D, desc.H, desc.W, desc.Do, desc.Ho, desc.Wo = d, h, w, d, h, w
    desc.k, desc.stride, desc.Ca = 3, 1, 4 * ca
    desc.xa_bs, desc.y_bs, desc.n_wptr = bs, cout * d * h * w, groups
    desc.pre, desc.pre_slope, desc.epi = 1, LEAK, 2
    return bool(L.load().xh_conv3d_supports_bcast(C.byref(desc)))


def init_fold_fwd(red_x, count, n, weights, eps=NORM_EPS):
    """InstanceNorm of the init blocks' 1x1 convs as a per-channel affine of their INPUT (xh_init_fold_fwd): red_x (n, M, 2) the fp64
    sums (sum x, sum x^2) of the M stored channels, weights: M tensors of B values (channel c = m * B + j computes w[m][j] * x_m +
    bias; the bias drops out of the norm).  Returns (sc, sh, rstd, ctr), each (n, M * B): IN(w x + b) = sc * x + sh; ctr = the fp32
    channel mean of x (the centre the data gradient's second sum is taken around)."""
    m, b = len(weights), weights[0].numel()
    sc, sh, rstd, ctr = (torch.empty((n, m * b), dtype=torch.float32, device=red_x.device) for _ in range(4))
    L.check(L.load().xh_init_fold_fwd(_stream(), _p(red_x), int(count), n, m, b, C.byref(_arr4([_f32(t.reshape(-1), "init weight") for t in weights])),
                                      float(eps), _p(sc), _p(sh), _p(rstd), _p(ctr)), "xh_init_fold_fwd")
    return sc, sh, rstd, ctr


def init_fold_bwd(red_x, count, n, weights, red_g, dws, eps=NORM_EPS):
    """dws[m][j] += d loss / d w[m][j] from the data gradient's sums red_g (n, M * B, 2) = (sum g, sum g (x - ctr)) (xh_init_fold_bwd)."""
    m, b = len(weights), weights[0].numel()
    L.check(L.load().xh_init_fold_bwd(_stream(), _p(red_x), int(count), n, m, b, C.byref(_arr4([_f32(t.reshape(-1), "init weight") for t in weights])),
                                      float(eps), _p(red_g), C.byref(_arr4([_f32(t.reshape(-1), "init dw") for t in dws]))), "xh_init_fold_bwd")


_C1_COLLECT = [None]


def conv1x1_collect():
    """The k = 1 conv3d calls from here to conv1x1_flush() are recorded instead of launched (their output tensors are returned as
    usual) and then issued as multi-problem launches of up to 4 (xh_conv1x1_multi): the VU-block convs of the four fusion levels."""
    _C1_COLLECT[0] = []


def conv1x1_flush():
    calls, _C1_COLLECT[0] = _C1_COLLECT[0] or [], None
    lib = L.load()
    for part in _chunks(calls):
        n = len(part)
        rc = 1
        if n > 1:
            descs = (C.POINTER(L.ConvDesc) * n)(*[C.pointer(c[0]) for c in part])
            ptrs = (C.POINTER(L.ConvPtrs) * n)(*[C.pointer(c[1]) for c in part])
            rc = lib.xh_conv1x1_multi(_stream(), n, descs, ptrs)
        if rc == 1:                               # not batchable (mixed epilogues / layouts): one by one
            for c in part:
                L.check(lib.xh_conv3d_fwd(_stream(), C.byref(c[0]), C.byref(c[1])), "xh_conv3d_fwd")
        else:
            L.check(rc, "xh_conv1x1_multi")


def conv3d_dgrad_s2(dy, weights, *, cin, in_spatial, groups=1, e=None, red=None):
    """Data gradient of the k3/s2/p1 conv: dy (N,Cout,Do,Ho,Wo) -> (N,cin,D,H,W)."""
    lib = L.load()
    n, cout, do, ho, wo, dy_bs = _vol(dy)
    _check_weights(weights, None, cin, cout, groups, 3, False, "conv3d_dgrad_s2")
    d, h, w = in_spatial
    out = new_like(dy, (n, cin, d, h, w))
    desc = L.ConvDesc()
    desc.dtype = _dt(dy)
    desc.arith = current_arith()
    desc.N, desc.Cin, desc.Cout, desc.groups = n, cin, cout, groups
    desc.D, desc.H, desc.W, desc.Do, desc.Ho, desc.Wo = d, h, w, do, ho, wo
    desc.k, desc.stride, desc.Ca = 3, 2, cout
    desc.xa_bs, desc.y_bs = dy_bs, _vol(out)[5]
    desc.n_wptr = len(weights)
    desc.epi = 1 if e is not None else 0
    ptrs = L.ConvPtrs()
    ptrs.xa, ptrs.y = _p(dy), _p(out)
    ptrs.w = _arr4([_f32(t, "conv weight") for t in weights])
    if e is not None:
        ea, eb = e[0], e[1]
        desc.Cea, desc.ea_bs = ea.shape[1], _vol(ea)[5]
        desc.eb_bs = _vol(eb)[5] if eb is not None else 0
        desc.e_slope = float(e[4])
        ptrs.ea, ptrs.eb, ptrs.e_sc, ptrs.e_sh, ptrs.red = _p(ea), _p(eb), _p(e[2]), _p(e[3]), _p(red)
        ptrs.fan, ptrs.fan_bytes = fan_block(dy.device)
    L.check(lib.xh_conv3d_dgrad_s2(_stream(), C.byref(desc), C.byref(ptrs)), "xh_conv3d_dgrad_s2")
    return out


# Weight gradients are off the critical path of backward: nothing downstream in the step reads them until the optimizer /
# the gradient all-reduce.  With overlap enabled they are launched on a second HIP stream that forks from the compute
# stream at the call (everything issued so far -- dY, the saved input, the zeroed gradient bucket -- is ordered before
# it) and is joined once, by join_wgrad_stream(), before the gradients are consumed.  The small-volume weight gradients
# are latency-bound (a few dozen workgroups): next to the data-gradient chain they cost almost nothing.  Inside a hipGraph
# capture the fork/join becomes a parallel branch of the graph.  Only calls whose targets are existing .grad buffers
# (functional._targets "direct" accumulation, e.g. parallel.FlatGrads) may overlap: a freshly allocated gradient that
# autograd consumes on the compute stream must be complete when backward() returns it.
_WG = {"on": False, "streams": {}, "keep": [], "forked": set(), "pending": [], "batch": 12, "defer": False, "deferred": [],
       "early": False, "early_done": False, "maxvol": 0}


def set_wgrad_overlap(enabled, batch=None):
    """Enable / disable launching weight-gradient kernels on a side stream (default off).  The caller must then call
    join_wgrad_stream() after backward() and before reading gradients (parallel.FlatGrads.all_reduce / .zero do).
    `batch`: weight-gradient calls are collected and forked to the side stream `batch` at a time (one cross-stream
    dependency per batch instead of one per call; a fork costs a few microseconds of its own)."""
    if not enabled:
        join_wgrad_stream()
    _WG["on"] = bool(enabled)
    if batch is not None:
        _WG["batch"] = max(1, int(batch))


def _flush_wgrads():
    pend = _WG["pending"]
    if not pend:
        return
    dev = pend[0][0].device
    st = _WG["streams"].get(dev)
    if st is None:
        st = _WG["streams"][dev] = torch.cuda.Stream(dev)
    st.wait_stream(torch.cuda.current_stream(dev))          # every dY / input of the batch has been issued by now
    _WG["forked"].add(dev)
    with torch.cuda.stream(st):
        for xa, xb, dy, dws, dbs, kw, ar in pend:
            with arith_scope(ar):                            # the mode of the backward that queued the call
                conv3d_wgrad(xa, xb, dy, dws, dbs, side=False, **kw)
    _WG["keep"].extend(pend)                                 # alive until the join: the allocator must not recycle them
    _WG["pending"] = []


def join_wgrad_stream():
    """Launches what is still pending and orders the compute stream behind every weight-gradient launch issued on the
    side stream since the last join."""
    if NB_PENDING:
        n_left = len(NB_PENDING)
        NB_PENDING.clear()
        raise RuntimeError(f"{n_left} norm-backward hand-over(s) were never taken: a gradient tensor of this backward pass was left "
                           "unwritten (functional.InLreluConv); ops.set_norm_bwd_fold(False) disables the hand-over")
    _flush_deferred()
    _flush_wgrads()
    for dev in list(_WG["forked"]):
        torch.cuda.current_stream(dev).wait_stream(_WG["streams"][dev])
    _WG["forked"].clear()
    _WG["keep"].clear()
    _WG["early_done"], _WG["maxvol"] = False, 0


def _wgrad_call(xa, xb, dy, dws, dbs, k, stride, groups, pre, bcast=0):
    """Marshals one weight-gradient problem: (desc, ptrs, dw[4], db[4], keep-alive list)."""
    lib = L.load()
    n, cout, do, ho, wo, dy_bs = _vol(dy)
    cin_l = xa.shape[1] * bcast if bcast else xa.shape[1] + (xb.shape[1] if xb is not None else 0)
    if bcast and (xb is not None or bcast != 4):
        raise ValueError("conv3d_wgrad: a broadcast input is one source, bcast = 4")
    _check_weights(dws, dbs, cin_l, cout, groups, k, False, "conv3d_wgrad")
    desc = _conv_desc(xa, xb, k, stride, groups, cout, len(dws), False, pre, ACT_NONE, LEAK, 0, None, 0, (do, ho, wo))
    if bcast:
        desc.bcast, desc.Cin, desc.Ca = bcast, cin_l, cin_l
    desc.ea_bs = dy_bs
    ptrs = L.ConvPtrs()
    ptrs.xa, ptrs.xb, ptrs.ea = _p(xa), _p(xb), _p(dy)
    ptrs.w = _arr4([_f32(t, "dw") for t in dws])     # only used for the non-NULL check
    if pre is not None:
        ptrs.pre_sc, ptrs.pre_sh = _p(pre[0]), _p(pre[1])
    dw = _arr4([_f32(t, "dw") for t in dws])
    db = _arr4([_f32(t, "db") for t in (dbs or [])])
    keep = [xa, xb, dy, pre, dws, dbs]
    need = lib.xh_conv3d_wgrad_workspace_bytes(C.byref(desc)) if _MFMA[0] else 0
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=xa.device)
        ptrs.ws, ptrs.ws_bytes = ws.data_ptr(), need
        keep.append(ws)
    return desc, ptrs, dw, db, keep


def conv3d_wgrad(xa, xb, dy, dws, dbs, *, k, stride=1, groups=1, pre=None, side=False, bcast=0):
    """Accumulates into the fp32 tensors dws (and dbs, may be None) the weight/bias gradients.  side=True: the targets are
    long-lived gradient buffers, so the call may be deferred and batched (set_wgrad_defer) or go to the weight-gradient
    stream (set_wgrad_overlap)."""
    if side and xa.is_cuda and _WG["defer"]:
        vol = dy.shape[2] * dy.shape[3] * dy.shape[4]
        if _WG["early"] and not _WG["early_done"] and _WG["deferred"] and vol * 64 <= _WG["maxvol"]:
            _flush_deferred_early(xa.device)
        _WG["deferred"].append(_wgrad_call(xa, xb, dy, dws, dbs, k, stride, groups, pre, bcast))
        _WG["maxvol"] = max(_WG["maxvol"], vol)
        return
    if side and _WG["on"] and xa.is_cuda:
        _WG["pending"].append((xa, xb, dy, dws, dbs, dict(k=k, stride=stride, groups=groups, pre=pre, bcast=bcast), current_arith()))
        if len(_WG["pending"]) >= _WG["batch"]:
            _flush_wgrads()
        return
    desc, ptrs, dw, db, keep = _wgrad_call(xa, xb, dy, dws, dbs, k, stride, groups, pre, bcast)
    if _WG["forked"]:
        _WG["keep"].append(keep)
    L.check(L.load().xh_conv3d_wgrad(_stream(), C.byref(desc), C.byref(ptrs), C.byref(dw), C.byref(db)), "xh_conv3d_wgrad")


def set_wgrad_defer(enabled):
    """Collect the weight-gradient calls of a backward pass and issue them together at join_wgrad_stream() through
    xh_conv3d_wgrad_batch: the k=3 MFMA problems share launches (default off; the caller must join before reading
    gradients -- parallel.FlatGrads.all_reduce / .zero do)."""
    if not enabled:
        join_wgrad_stream()
    _WG["defer"] = bool(enabled)


def drop_deferred_wgrads():
    """Forgets the queued weight-gradient calls without launching them (error paths: a backward that raised, a failed capture)."""
    _WG["deferred"] = []
    _WG["pending"] = []
    _WG["early_done"], _WG["maxvol"] = False, 0


def _batch_call(calls):
    n = len(calls)
    descs = (C.POINTER(L.ConvDesc) * n)(*[C.pointer(c[0]) for c in calls])
    ptrs = (C.POINTER(L.ConvPtrs) * n)(*[C.pointer(c[1]) for c in calls])
    dws = ((C.c_void_p * L.MAX_WPTR) * n)()
    dbs = ((C.c_void_p * L.MAX_WPTR) * n)()
    for i, c in enumerate(calls):
        for j in range(L.MAX_WPTR):
            dws[i][j] = c[2][j]
            dbs[i][j] = c[3][j]
    L.check(L.load().xh_conv3d_wgrad_batch(_stream(), n, descs, ptrs, dws, dbs), "xh_conv3d_wgrad_batch")


def set_wgrad_early_flush(enabled):
    """Deferred weight gradients (set_wgrad_defer): when the backward pass first descends two resolution levels below the largest
    volume queued so far, launch what is queued -- the decoder's full- and half-resolution problems, large launches -- on the side
    stream instead of keeping it for the end.  They then run beside the ~80 launches of the deep levels (32^3 and below, the mLSTM
    block, PoE: grids of 4 - 512 workgroups that leave most of the chip idle); the rest of the queue is launched at the end of the
    pass as before.  One fork, one join (capture-safe).  Default OFF: MEASURED SLOWER on MI355X (hipGraph replay of the 128^3 step:
    3.86 ms without, 4.05 ms with) -- the weight-gradient kernels are persistent workgroups sized to hold every CU for their whole
    run, so the deep levels' short launches, which ARE the critical path there, queue behind them for wave slots and LDS.  Kept as an
    A/B knob (bench.py --wgrad-early-flush)."""
    _WG["early"] = bool(enabled)


def _flush_deferred_early(dev):
    calls = _WG["deferred"]
    _WG["deferred"] = []
    _WG["early_done"] = True
    st = _WG["streams"].get(dev)
    if st is None:
        st = _WG["streams"][dev] = torch.cuda.Stream(dev)
    st.wait_stream(torch.cuda.current_stream(dev))          # every dY / input of the queue has been issued by now
    _WG["forked"].add(dev)
    with torch.cuda.stream(st):
        _batch_call(calls)
    _WG["keep"].append(calls)                                # alive until the join: the allocator must not recycle them


def set_wgrad_flush_streams(n):
    """The deferred weight gradients are independent of each other and of everything after them: with n > 1 the flush
    forks n - 1 extra HIP streams, deals the problems to them by kernel family (so that the batched k=3 launches stay
    together) and joins once -- the small latency-bound launches run side by side.  Capture-safe (fork / join through
    stream waits).  Default 1: MEASURED SLOWER on MI355X (hipGraph replay of the 128^3 step: 6.47 ms with 1 stream, 6.80
    with 2, 7.82 with 3, 9.14 with 4) -- every launch is already sized to fill the chip, so concurrent launches only contend;
    kept as an A/B knob (bench.py --wgrad-flush-streams)."""
    _WG["flush_streams"] = max(1, int(n))


def _flush_deferred():
    calls = _WG["deferred"]
    if not calls:
        return
    _WG["deferred"] = []
    ns = _WG.get("flush_streams", 1)
    if ns <= 1 or len(calls) < 2 * ns:
        return _batch_call(calls)
    dev = torch.cuda.current_device()
    pool = _WG.setdefault("flush_pool", {}).setdefault(dev, [])
    while len(pool) < ns - 1:
        pool.append(torch.cuda.Stream(dev))
    # group 0 (current stream): the k = 3 problems (batched launches); the others round-robin over the side streams
    groups = [[] for _ in range(ns)]
    rest = 0
    for c in calls:
        if c[0].k == 3 and c[0].stride == 1 and c[0].Cin // c[0].groups >= 4:
            groups[0].append(c)
        else:
            groups[1 + rest % (ns - 1)].append(c)
            rest += 1
    cur = torch.cuda.current_stream(dev)
    for i in range(1, ns):
        if groups[i]:
            pool[i - 1].wait_stream(cur)
            with torch.cuda.stream(pool[i - 1]):
                _batch_call(groups[i])
    if groups[0]:
        _batch_call(groups[0])
    for i in range(1, ns):
        if groups[i]:
            cur.wait_stream(pool[i - 1])


# ----------------------------------------------------------------------------------------------- norms
def moments(x, red, c0=0):
    """red[:, c0:c0+C] += (sum x, sum x^2)."""
    n, c, d, h, w, bs = _vol(x)
    sl = red[:, c0:c0 + c]
    L.check(L.load().xh_moments(_stream(), _dt(x), _p(x), bs, n, c, d * h * w, sl.data_ptr(), red.stride(0)), "xh_moments")


def moments2(xa, xb, red):
    """red[:, :ca+cb] += (sum, sum of squares) of the virtual concat (xa | xb), one launch."""
    n, ca, d, h, w, bsa = _vol(xa)
    cb, bsb = xb.shape[1], _vol(xb)[5]
    L.check(L.load().xh_moments2(_stream(), _dt(xa), _p(xa), bsa, ca, _p(xb), bsb, cb, n, d * h * w, red.data_ptr(), red.stride(0)),
            "xh_moments2")


def norm_finalize(mode, red, n, c, count, *, gs=1, gamma=None, beta=None, running_mean=None, running_var=None,
                  steps=1, device=None):
    dev = red.device if red is not None else device
    sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=dev) for _ in range(4))
    L.check(L.load().xh_norm_finalize(_stream(), mode, _p(red), n, c, count, gs, NORM_EPS, _p(gamma), _p(beta),
                                      _p(running_mean), _p(running_var), steps, _p(sc), _p(sh), _p(mean), _p(rstd)),
            "xh_norm_finalize")
    return sc, sh, mean, rstd


def affine_act(x, sc, sh, act, slope=LEAK, out=None):
    n, c, d, h, w, bs = _vol(x)
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    L.check(L.load().xh_affine_act(_stream(), _dt(x), _p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w, _p(sc), _p(sh),
                                   act, slope), "xh_affine_act")
    return out


def in_affine_act(x, red, act, slope=LEAK):
    """InstanceNorm (from the raw channel sums `red`) + activation in one launch; returns y, sc, sh, mean, rstd."""
    n, c, d, h, w, bs = _vol(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
    L.check(L.load().xh_in_affine_act(_stream(), _dt(x), _p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w, _p(red), act, slope,
                                      _p(sc), _p(sh), _p(mean), _p(rstd)), "xh_in_affine_act")
    return out, sc, sh, mean, rstd


def bn_affine_act(mode, x, red, act, *, gamma=None, beta=None, running_mean=None, running_var=None, steps=1, slope=LEAK):
    """BatchNorm finalisation (train: batch statistics from `red`, running statistics updated; eval: running statistics) +
    activation in one launch; returns y, sc, sh, mean, rstd."""
    n, c, d, h, w, bs = _vol(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
    L.check(L.load().xh_bn_affine_act(_stream(), _dt(x), mode, _p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w, _p(red), NORM_EPS,
                                      _p(gamma), _p(beta), _p(running_mean), _p(running_var), steps, act, slope, _p(sc), _p(sh),
                                      _p(mean), _p(rstd)), "xh_bn_affine_act")
    return out, sc, sh, mean, rstd


def bn_affine_act2(mode, x, red, act, chalf, *, gammas, betas, running_means, running_vars, steps=1, slope=LEAK):
    """bn_affine_act for TWO BatchNorm modules over the channel halves of one tensor (xh_bn_affine_act2): channels [0, chalf) take
    gammas[0] / betas[0] / ..., the rest the second set."""
    n, c, d, h, w, bs = _vol(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
    L.check(L.load().xh_bn_affine_act2(_stream(), _dt(x), mode, _p(x), bs, _p(out), _vol(out)[5], n, c, int(chalf), d * h * w, _p(red),
                                       NORM_EPS, _p(gammas[0]), _p(betas[0]), _p(running_means[0]), _p(running_vars[0]), _p(gammas[1]),
                                       _p(betas[1]), _p(running_means[1]), _p(running_vars[1]), steps, act, slope, _p(sc), _p(sh),
                                       _p(mean), _p(rstd)), "xh_bn_affine_act2")
    return out, sc, sh, mean, rstd


def norm_bwd_fused2(mode, dy, x, red, mean, rstd, chalf, *, gammas, dgammas, dbetas):
    """norm_bwd_fused (BatchNorm modes) for two modules over the channel halves of one tensor (xh_norm_bwd_fused2)."""
    n, c, d, h, w, bs = _vol(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    L.check(L.load().xh_norm_bwd_fused2(_stream(), _dt(x), mode, _p(dy), _vol(dy)[5], _p(x), bs, _p(out), _vol(out)[5], n, c, int(chalf),
                                        d * h * w, _p(red), _p(gammas[0]), _p(gammas[1]), _p(mean), _p(rstd), _p(dgammas[0]),
                                        _p(dbetas[0]), _p(dgammas[1]), _p(dbetas[1])), "xh_norm_bwd_fused2")
    return out


def act_bwd_reduce(dy, x, sc, sh, slope):
    n, c, d, h, w, bs = _vol(x)
    red = zeros_red(x, n, c)
    L.check(L.load().xh_act_bwd_reduce(_stream(), _dt(x), _p(dy), _vol(dy)[5], _p(x), bs, n, c, d * h * w, _p(sc), _p(sh),
                                       slope, _p(red)), "xh_act_bwd_reduce")
    return red


def norm_bwd_coef(mode, red, count, mean, rstd, *, gs=1, gamma=None, dgamma=None, dbeta=None):
    n, c = mean.shape
    A, B, Cc = (torch.empty((n, c), dtype=torch.float32, device=mean.device) for _ in range(3))
    L.check(L.load().xh_norm_bwd_coef(_stream(), mode, _p(red), n, c, count, gs, _p(gamma), _p(mean), _p(rstd), _p(A),
                                      _p(B), _p(Cc), _p(dgamma), _p(dbeta)), "xh_norm_bwd_coef")
    return A, B, Cc


def norm_bwd_apply(dy, x, coef, *, have_g, sc=None, sh=None, slope=LEAK, out=None, accumulate=False, c0=0):
    """dx (+)= A*g + Cc*x + B over x's channels; dy/coef may be wider: channel window starts at c0."""
    n, c, d, h, w, bs = _vol(x)
    A, B, Cc = (t[:, c0:c0 + c].contiguous() if t.shape[1] != c else t for t in coef)
    if sc is not None and sc.shape[1] != c:
        sc, sh = sc[:, c0:c0 + c].contiguous(), sh[:, c0:c0 + c].contiguous()
    dyv = dy[:, c0:c0 + c] if dy.shape[1] != c else dy
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    L.check(L.load().xh_norm_bwd_apply(_stream(), _dt(x), _p(dyv), _vol(dyv)[5], _p(x), bs, _p(out), _vol(out)[5], n, c,
                                       d * h * w, _p(A), _p(B), _p(Cc), int(have_g), _p(sc), _p(sh), slope,
                                       int(accumulate)), "xh_norm_bwd_apply")
    return out


def norm_bwd_fused(mode, dy, x, red, mean, rstd, *, gs=1, gamma=None, dgamma=None, dbeta=None):
    """norm_bwd_coef + norm_bwd_apply(have_g=True) of a BatchNorm / GroupNorm in one launch (count = x's volume)."""
    n, c, d, h, w, bs = _vol(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    L.check(L.load().xh_norm_bwd_fused(_stream(), _dt(x), mode, _p(dy), _vol(dy)[5], _p(x), bs, _p(out), _vol(out)[5], n, c,
                                       d * h * w, _p(red), gs, _p(gamma), _p(mean), _p(rstd), _p(dgamma), _p(dbeta)),
            "xh_norm_bwd_fused")
    return out


def in_bwd_apply(dy, x, red, mean, rstd, *, have_g, sc=None, sh=None, slope=LEAK, c0=0, acc=None, out=None):
    """InstanceNorm backward of x's channels (window starting at c0 of dy / the statistics) in one launch.  `acc`: an existing
    gradient buffer of x's shape to ADD the result to (functional.GradSlot) instead of a new tensor; `out`: a tensor to write."""
    n, c, d, h, w, bs = _vol(x)
    rs = mean.shape[1]
    dyv = dy[:, c0:c0 + c] if dy.shape[1] != c else dy
    if out is None:
        out = acc if acc is not None else torch.empty_like(x, memory_format=torch.contiguous_format)
    off = lambda t, scale=1: None if t is None else t.data_ptr() + c0 * scale * t.element_size()
    L.check(L.load().xh_in_bwd_apply(_stream(), _dt(x), _p(dyv), _vol(dyv)[5], _p(x), bs, _p(out), _vol(out)[5], n, c,
                                     d * h * w, off(red, 2), off(mean), off(rstd), rs, int(have_g), off(sc), off(sh), slope,
                                     int(acc is not None)), "xh_in_bwd_apply")
    return out


def in_bwd_apply2(dy, xa, xb, red, mean, rstd, acc_a=None, acc_b=None, out_a=None, out_b=None):
    """InstanceNorm backward of the virtual concat (xa | xb) from its full-width gradient g = dy, one launch.  acc_a / acc_b:
    existing gradient buffers to add the respective half to; out_a / out_b: tensors to WRITE the half to (else new ones)."""
    n, ca, d, h, w, bsa = _vol(xa)
    cb, bsb = xb.shape[1], _vol(xb)[5]
    da = acc_a if acc_a is not None else out_a if out_a is not None else torch.empty_like(xa, memory_format=torch.contiguous_format)
    db = acc_b if acc_b is not None else out_b if out_b is not None else torch.empty_like(xb, memory_format=torch.contiguous_format)
    L.check(L.load().xh_in_bwd_apply2(_stream(), _dt(xa), _p(dy), _vol(dy)[5], _p(xa), bsa, _p(da), _vol(da)[5], ca, _p(xb), bsb,
                                      _p(db), _vol(db)[5], cb, n, d * h * w, _p(red), _p(mean), _p(rstd),
                                      int(acc_a is not None) | (int(acc_b is not None) << 1)), "xh_in_bwd_apply2")
    return da, db


# ----------------------------------------------------------------------------------------------- resampling
def maxpool2(x):
    n, c, d, h, w, _ = _vol(x)
    x = x.contiguous()
    y = new_like(x, (n, c, d // 2, h // 2, w // 2))
    L.check(L.load().xh_maxpool2_fwd(_stream(), _dt(x), _p(x), _p(y), n * c, d, h, w), "xh_maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy, acc=None):
    n, c, d, h, w, _ = _vol(x)
    x, dy = x.contiguous(), dy.contiguous()
    if acc is not None and (not acc.is_contiguous() or acc.shape != x.shape or acc.dtype != x.dtype):
        raise ValueError("maxpool2_bwd: acc must be a contiguous tensor of x's shape and type")   # (a fresh dx would drop this share)
    dx = acc if acc is not None else torch.empty_like(x)
    L.check(L.load().xh_maxpool2_bwd(_stream(), _dt(x), _p(x), _p(dy), _p(dx), n * c, d, h, w, int(dx is acc)), "xh_maxpool2_bwd")
    return dx


def upsample(x, size, out=None):
    n, c, d, h, w, bs = _vol(x)
    do, ho, wo = size
    if out is None:
        out = new_like(x, (n, c, do, ho, wo))
    L.check(L.load().xh_upsample_trilinear_fwd(_stream(), _dt(x), _p(x), bs, _p(out), _vol(out)[5], n, c, d, h, w, do, ho, wo),
            "xh_upsample_trilinear_fwd")
    return out


def upsample_bwd(dy, in_size):
    n, c, do, ho, wo, bs = _vol(dy)
    d, h, w = in_size
    dx = new_like(dy, (n, c, d, h, w))
    L.check(L.load().xh_upsample_trilinear_bwd(_stream(), _dt(dy), _p(dy), bs, _p(dx), _vol(dx)[5], n, c, d, h, w, do, ho, wo, 0),
            "xh_upsample_trilinear_bwd")
    return dx


def upsample2x_in_act(x, red, slope=LEAK):
    """up2x(leaky(InstanceNorm(x))) with the norm finalised from the raw channel sums `red` in the same launch
    (xh_upsample2x_in_act_fwd); returns (y, sc, sh, mean, rstd), or None when the exact-2x kernel does not take the layout."""
    n, c, d, h, w, bs = _vol(x)
    out = new_like(x, (n, c, 2 * d, 2 * h, 2 * w))
    sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
    rc = L.load().xh_upsample2x_in_act_fwd(_stream(), _dt(x), _p(x), bs, _p(out), _vol(out)[5], n, c, d, h, w, _p(red), slope,
                                           _p(sc), _p(sh), _p(mean), _p(rstd))
    if rc == 1:
        return None
    L.check(rc, "xh_upsample2x_in_act_fwd")
    return out, sc, sh, mean, rstd


def upsample2x_bwd_act_reduce(dy, y0, sc, sh, slope=LEAK):
    """(dx, red): the adjoint of the exact-2x upsampling and the activation-masked sums of act_bwd_reduce over it, one launch
    (xh_upsample2x_bwd_act_reduce); None when the exact-2x kernel does not take the layout."""
    n, c, d, h, w, bs = _vol(y0)
    if tuple(dy.shape[2:]) != (2 * d, 2 * h, 2 * w):
        return None
    dx = new_like(dy, (n, c, d, h, w))
    red = zeros_red(y0, n, c)
    rc = L.load().xh_upsample2x_bwd_act_reduce(_stream(), _dt(dy), _p(dy), _vol(dy)[5], _p(dx), _vol(dx)[5], n, c, d, h, w, _p(y0), bs,
                                               _p(sc), _p(sh), slope, _p(red))
    if rc == 1:
        return None
    L.check(rc, "xh_upsample2x_bwd_act_reduce")
    return dx, red


# ---- multi-problem launches of the latent path's passes (include/xlstm_hved.h: xh_*_multi): the same pass over up to 4 tensors of
# different sizes in ONE launch.  Each returns a list with one entry per problem, shaped like the single-problem op's result.
def _chunks(seq):
    return [seq[i:i + L.MULTI_MAX] for i in range(0, len(seq), L.MULTI_MAX)]


def in_affine_act_multi(xs, reds, act, slope=LEAK):
    outs = []
    for part in _chunks(list(zip(xs, reds))):
        arr = (L.InAffineActArgs * len(part))()
        res = []
        for i, (x, red) in enumerate(part):
            n, c, d, h, w, bs = _vol(x)
            out = torch.empty_like(x, memory_format=torch.contiguous_format)
            sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
            arr[i] = L.InAffineActArgs(_p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w, _p(red), act, slope, _p(sc), _p(sh), _p(mean), _p(rstd))
            res.append((out, sc, sh, mean, rstd))
        L.check(L.load().xh_in_affine_act_multi(_stream(), _dt(part[0][0]), len(part), C.addressof(arr)), "xh_in_affine_act_multi")
        outs += res
    return outs


def act_bwd_reduce_multi(dys, xs, scs, shs, slope):
    outs = []
    for part in _chunks(list(zip(dys, xs, scs, shs))):
        arr = (L.ActBwdReduceArgs * len(part))()
        res = []
        for i, (dy, x, sc, sh) in enumerate(part):
            n, c, d, h, w, bs = _vol(x)
            red = zeros_red(x, n, c)
            arr[i] = L.ActBwdReduceArgs(_p(dy), _vol(dy)[5], _p(x), bs, n, c, d * h * w, _p(sc), _p(sh), slope, _p(red))
            res.append(red)
        L.check(L.load().xh_act_bwd_reduce_multi(_stream(), _dt(part[0][1]), len(part), C.addressof(arr)), "xh_act_bwd_reduce_multi")
        outs += res
    return outs


def in_bwd_apply_multi(dys, xs, reds, means, rstds, *, have_g, scs=None, shs=None, slope=LEAK):
    outs = []
    scs = scs if scs is not None else [None] * len(xs)
    shs = shs if shs is not None else [None] * len(xs)
    for part in _chunks(list(zip(dys, xs, reds, means, rstds, scs, shs))):
        arr = (L.InBwdApplyArgs * len(part))()
        res = []
        for i, (dy, x, red, mean, rstd, sc, sh) in enumerate(part):
            n, c, d, h, w, bs = _vol(x)
            out = torch.empty_like(x, memory_format=torch.contiguous_format)
            arr[i] = L.InBwdApplyArgs(_p(dy), _vol(dy)[5], _p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w, _p(red), _p(mean), _p(rstd),
                                      mean.shape[1], int(have_g), _p(sc), _p(sh), slope, 0)
            res.append(out)
        L.check(L.load().xh_in_bwd_apply_multi(_stream(), _dt(part[0][1]), len(part), C.addressof(arr)), "xh_in_bwd_apply_multi")
        outs += res
    return outs


def upsample2x_in_act_multi(xs, reds, slope=LEAK):
    """None when a problem's layout is not taken by the exact-2x kernel (the caller then runs the single-problem ops)."""
    outs = []
    for part in _chunks(list(zip(xs, reds))):
        arr = (L.Upsample2xInActArgs * len(part))()
        res = []
        for i, (x, red) in enumerate(part):
            n, c, d, h, w, bs = _vol(x)
            out = new_like(x, (n, c, 2 * d, 2 * h, 2 * w))
            sc, sh, mean, rstd = (torch.empty((n, c), dtype=torch.float32, device=x.device) for _ in range(4))
            arr[i] = L.Upsample2xInActArgs(_p(x), bs, _p(out), _vol(out)[5], n, c, d, h, w, _p(red), slope, _p(sc), _p(sh), _p(mean), _p(rstd))
            res.append((out, sc, sh, mean, rstd))
        rc = L.load().xh_upsample2x_in_act_multi(_stream(), _dt(part[0][0]), len(part), C.addressof(arr))
        if rc == 1:
            return None
        L.check(rc, "xh_upsample2x_in_act_multi")
        outs += res
    return outs


def upsample2x_bwd_act_reduce_multi(dys, y0s, scs, shs, slope=LEAK):
    outs = []
    for part in _chunks(list(zip(dys, y0s, scs, shs))):
        arr = (L.Upsample2xBwdArgs * len(part))()
        res = []
        for i, (dy, y0, sc, sh) in enumerate(part):
            n, c, d, h, w, bs = _vol(y0)
            if tuple(dy.shape[2:]) != (2 * d, 2 * h, 2 * w):
                return None
            dx = new_like(dy, (n, c, d, h, w))
            red = zeros_red(y0, n, c)
            arr[i] = L.Upsample2xBwdArgs(_p(dy), _vol(dy)[5], _p(dx), _vol(dx)[5], n, c, d, h, w, _p(y0), bs, _p(sc), _p(sh), slope, _p(red))
            res.append((dx, red))
        rc = L.load().xh_upsample2x_bwd_act_reduce_multi(_stream(), _dt(part[0][0]), len(part), C.addressof(arr))
        if rc == 1:
            return None
        L.check(rc, "xh_upsample2x_bwd_act_reduce_multi")
        outs += res
    return outs


def add(a, b, out=None):
    """out = a + b (b None: copy) for NCDHW-blocked tensors."""
    n, c, d, h, w, a_bs = _vol(a)
    if out is None:
        out = new_like(a, (n, c, d, h, w))
    L.check(L.load().xh_add(_stream(), _dt(a), _p(a), a_bs, _p(b), _vol(b)[5] if b is not None else 0, _p(out), _vol(out)[5],
                            n, c * d * h * w), "xh_add")
    return out


def act_bwd(dy, y, act):
    dy, y = dy.contiguous(), y.contiguous()
    dx = torch.empty_like(y)
    L.check(L.load().xh_act_bwd(_stream(), _dt(y), _p(dy), _p(y), _p(dx), y.numel(), act), "xh_act_bwd")
    return dx


# ----------------------------------------------------------------------------------------------- PoE
def poe_fwd(feat, keep, eps, L_, mask_mu):
    n, c, d, h, w, _ = _vol(feat)
    feat = feat.contiguous()
    z = new_like(feat, (n, L_, d, h, w))
    mu = new_like(feat, (n, 5, L_, d, h, w))
    lv = new_like(feat, (n, 5, L_, d, h, w))
    L.check(L.load().xh_poe_fwd(_stream(), _dt(feat), _p(feat), _p(keep), _p(eps), _p(z), _p(mu), _p(lv), n, L_, d * h * w,
                                int(mask_mu)), "xh_poe_fwd")
    return z, mu, lv


def poe_bwd(feat, keep, eps, dz, dmu, dlv, L_, mask_mu):
    n, c, d, h, w, _ = _vol(feat)
    dfeat = torch.empty_like(feat)
    L.check(L.load().xh_poe_bwd(_stream(), _dt(feat), _p(feat), _p(keep), _p(eps), _p(dz), _p(dmu), _p(dlv), _p(dfeat), n, L_,
                                d * h * w, int(mask_mu)), "xh_poe_bwd")
    return dfeat


POE_MAX = 8


RNG_WORDS = 16 * (2 + 32)          # == XH_RNG_WORDS: {seed, counter, ..., ticket lines}


def rng_state(seed, device):
    """A fresh generator state for poe_fwd_multi(rng=...): int64[RNG_WORDS] = {seed, draw counter 0, zeros (ticket words)}."""
    st = torch.zeros(RNG_WORDS, dtype=torch.int64)
    st[0] = int(seed)
    return st.to(device)


def _rng_words(t, n, what):
    if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
        raise ValueError(f"{what}: {n} contiguous int64 device words")
    return t


def poe_fwd_multi(feats, keep, epss, Ls, mask_mu, rng=None):
    """poe_fwd for several latent levels in one launch (xh_poe_multi); returns [(z, mu, lv)] per level.
    rng = (state, used): the levels whose eps is None draw their noise IN the kernel (Philox4x32-10, fp32) from the generator
    `state` (ops.rng_state: int64[RNG_WORDS] device tensor {seed, counter, ticket words}; the launch advances the counter) and leave {counter, seed} of the
    draw in `used` (int64[2]) for poe_bwd_multi.  Without rng, eps None means the posterior mean."""
    jobs = (L.PoeJob * len(feats))()
    outs, keepalive = [], []
    for lvl, (j, feat, eps, L_) in enumerate(zip(jobs, feats, epss, Ls)):
        n, c, d, h, w, _ = _vol(feat)
        z = new_like(feat, (n, L_, d, h, w))
        mu = new_like(feat, (n, 5, L_, d, h, w))
        lv = new_like(feat, (n, 5, L_, d, h, w))
        j.feat, j.keep, j.eps, j.z, j.mu_stack, j.lv_stack = _p(feat), _p(keep), _p(eps), _p(z), _p(mu), _p(lv)
        j.dhw, j.N, j.L, j.mask_mu = d * h * w, n, L_, int(mask_mu)
        if rng is not None and eps is None:
            j.rng_used, j.rng_stream = _p(_rng_words(rng[1], 2, "rng used")), lvl
        outs.append((z, mu, lv))
    state = _p(_rng_words(rng[0], RNG_WORDS, "rng state")) if rng is not None else None
    L.check(L.load().xh_poe_multi(_stream(), _dt(feats[0]), 0, len(feats), C.cast(jobs, C.c_void_p), state), "xh_poe_multi")
    return outs


def poe_bwd_multi(feats, keep, epss, dzs, dmus, dlvs, Ls, mask_mu, rng_used=None):
    """poe_bwd for several latent levels in one launch; returns [dfeat] per level.  rng_used: the `used` words of the forward
    (levels with eps None regenerate their noise from them)."""
    jobs = (L.PoeJob * len(feats))()
    outs = []
    for lvl, (j, feat, eps, dz, dmu, dlv, L_) in enumerate(zip(jobs, feats, epss, dzs, dmus, dlvs, Ls)):
        n, c, d, h, w, _ = _vol(feat)
        dfeat = torch.empty_like(feat)
        j.feat, j.keep, j.eps, j.dz, j.dmu_stack, j.dlv_stack, j.dfeat = _p(feat), _p(keep), _p(eps), _p(dz), _p(dmu), _p(dlv), _p(dfeat)
        j.dhw, j.N, j.L, j.mask_mu = d * h * w, n, L_, int(mask_mu)
        if rng_used is not None and eps is None:
            j.rng_used, j.rng_stream = _p(_rng_words(rng_used, 2, "rng used")), lvl
        outs.append(dfeat)
    L.check(L.load().xh_poe_multi(_stream(), _dt(feats[0]), 1, len(feats), C.cast(jobs, C.c_void_p), None), "xh_poe_multi")
    return outs


def philox_normal(seed, counter, stream, n, device, raw=False):
    """The noise xh_poe_multi draws for (seed, counter, level = stream): n fp32 normals, or (n, 4) raw Philox words (int32 bits)."""
    out = torch.empty((n, 4), dtype=torch.int32, device=device) if raw else torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.load().xh_philox_normal(_stream(), int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1), int(stream), _p(out), n, int(raw)),
            "xh_philox_normal")
    return out


# ----------------------------------------------------------------------------------------------- attention glue
def channel_pool(x, out):
    """out (a 2-channel slice) = [max_c x, mean_c x]."""
    n, c, d, h, w, bs = _vol(x)
    L.check(L.load().xh_channel_pool_fwd(_stream(), _dt(x), _p(x), bs, _p(out), _vol(out)[5], n, c, d * h * w), "xh_channel_pool_fwd")


def channel_pool_bwd(x, dy, acc=None):
    n, c, d, h, w, bs = _vol(x)
    dx = acc if acc is not None else new_like(x, (n, c, d, h, w))
    L.check(L.load().xh_channel_pool_bwd(_stream(), _dt(x), _p(x), bs, _p(dy), _vol(dy)[5], _p(dx), _vol(dx)[5], n, c, d * h * w,
                                         int(acc is not None)), "xh_channel_pool_bwd")
    return dx


def gate(x, s, out=None):
    n, c, d, h, w, bs = _vol(x)
    if out is None:
        out = new_like(x, (n, c, d, h, w))
    L.check(L.load().xh_gate_fwd(_stream(), _dt(x), _p(x), bs, _p(s), _vol(s)[5], _p(out), _vol(out)[5], n, c, d * h * w), "xh_gate_fwd")
    return out


def gate_bwd(x, s, dy, ds_out=None, acc=None):
    n, c, d, h, w, bs = _vol(x)
    dx = acc if acc is not None else new_like(x, (n, c, d, h, w))
    if ds_out is None:
        ds_out = new_like(x, (n, 1, d, h, w))
    L.check(L.load().xh_gate_bwd(_stream(), _dt(x), _p(x), bs, _p(s), _vol(s)[5], _p(dy), _vol(dy)[5], _p(dx), _vol(dx)[5],
                                 _p(ds_out), _vol(ds_out)[5], n, c, d * h * w, int(acc is not None), 0), "xh_gate_bwd")
    return dx, ds_out


_GMP = [os.environ.get("XH_NO_GATE_MAXPOOL", "") == ""]          # A/B switch (measurements): the fused gate + max-pool pass


# AttenModule2's pooled / gated pairs in one launch each (csrc/eltwise.hip, "AttenModule2 pairs")
def channel_pool2(a, b):
    n, ca, d, h, w, bsa = _vol(a)
    cb, bsb = b.shape[1], _vol(b)[5]
    y = new_like(a, (n, 4, d, h, w))
    L.check(L.load().xh_channel_pool2_fwd(_stream(), _dt(a), _p(a), bsa, ca, _p(b), bsb, cb, _p(y), _vol(y)[5], n, d * h * w),
            "xh_channel_pool2_fwd")
    return y


def channel_pool2_bwd(a, b, dy, acc_a=None, acc_b=None, out_a=None):
    n, ca, d, h, w, bsa = _vol(a)
    cb, bsb = b.shape[1], _vol(b)[5]
    da = acc_a if acc_a is not None else out_a if out_a is not None else new_like(a, (n, ca, d, h, w))
    db = acc_b if acc_b is not None else new_like(b, (n, cb, d, h, w))
    L.check(L.load().xh_channel_pool2_bwd(_stream(), _dt(a), _p(a), bsa, ca, _p(b), bsb, cb, _p(dy), _vol(dy)[5], _p(da), _vol(da)[5],
                                          int(acc_a is not None), _p(db), _vol(db)[5], int(acc_b is not None), n, d * h * w),
            "xh_channel_pool2_bwd")
    return da, db


def gate2(a, b, E, red=None):
    """cat[a * (1 + E[:, 0]), b * (1 + E[:, 1])]; with `red` (zeroed (n, ca + cb, 2) fp64) the channel sums of the output too."""
    n, ca, d, h, w, bsa = _vol(a)
    cb, bsb = b.shape[1], _vol(b)[5]
    y = new_like(a, (n, ca + cb, d, h, w))
    L.check(L.load().xh_gate2_fwd(_stream(), _dt(a), _p(a), bsa, ca, _p(b), bsb, cb, _p(E), _vol(E)[5], _p(y), _vol(y)[5], n, d * h * w,
                                  _p(red)), "xh_gate2_fwd")
    return y


def gate2_bwd(a, b, E, dy, acc_a=None, acc_b=None, sig_bwd=False, out_a=None):
    """sig_bwd: E is a sigmoid's output; dE comes back as the gradient of the pre-activation (dE * E * (1 - E)).  out_a: where to
    WRITE a's gradient when there is no buffer to add to (else a new tensor)."""
    n, ca, d, h, w, bsa = _vol(a)
    cb, bsb = b.shape[1], _vol(b)[5]
    da = acc_a if acc_a is not None else out_a if out_a is not None else new_like(a, (n, ca, d, h, w))
    db = acc_b if acc_b is not None else new_like(b, (n, cb, d, h, w))
    dE = torch.empty_like(E, memory_format=torch.contiguous_format)
    L.check(L.load().xh_gate2_bwd(_stream(), _dt(a), _p(a), bsa, ca, _p(b), bsb, cb, _p(E), _vol(E)[5], _p(dy), _vol(dy)[5], _p(da),
                                  _vol(da)[5], int(acc_a is not None), _p(db), _vol(db)[5], int(acc_b is not None), _p(dE), _vol(dE)[5],
                                  n, d * h * w, int(bool(sig_bwd))), "xh_gate2_bwd")
    return da, db, dE


def gate_maxpool_ok(x, s):
    """Shapes the fused gate + max-pool kernels take (else: gate, maxpool2 and moments one after the other)."""
    n, c, d, h, w, bs = _vol(x)
    if not _GMP[0]:
        return False
    if s is None:                       # max-pool + channel sums, no gate
        return d % 2 == 0 and h % 2 == 0 and w % 8 == 0 and bs % 8 == 0
    return d % 2 == 0 and h % 2 == 0 and w % 8 == 0 and bs % 8 == 0 and _vol(s)[5] % 8 == 0 and s.shape[1] == 1


def gate_maxpool(x, s, red=None, gated=0):
    """maxpool2(x * (1 + s)) in one pass (s None: maxpool2(x)); with `red` (zeroed (n, c, 2) fp64) the channel sums of the pooled
    output too.  gated: the gate applies to the first `gated` channels only (0: all)."""
    n, c, d, h, w, bs = _vol(x)
    y = new_like(x, (n, c, d // 2, h // 2, w // 2))
    L.check(L.load().xh_gate_maxpool_fwd(_stream(), _dt(x), _p(x), bs, _p(s), _vol(s)[5] if s is not None else 0, _p(y), _vol(y)[5], n, c, d, h, w,
                                         _p(red), int(gated)),
            "xh_gate_maxpool_fwd")
    return y


def gate_maxpool_bwd(x, s, dy, acc=None, gated=0, out=None):
    n, c, d, h, w, bs = _vol(x)
    dy = dy.contiguous()
    dx = acc if acc is not None else out if out is not None else new_like(x, (n, c, d, h, w))
    ds = new_like(x, (n, 1, d, h, w))
    L.check(L.load().xh_gate_maxpool_bwd(_stream(), _dt(x), _p(x), bs, _p(s), _vol(s)[5], _p(dy), _vol(dy)[5], _p(dx), _vol(dx)[5],
                                         _p(ds), _vol(ds)[5], n, c, d, h, w, int(acc is not None), int(gated)), "xh_gate_maxpool_bwd")
    return dx, ds


def duse_gate(x, ch, sp, red=None):
    """u = x * (1 + ch + sp); with `red` (zeroed (n, c, 2) fp64) the channel sums of u are left there in the same pass."""
    n, c, d, h, w, bs = _vol(x)
    u = new_like(x, (n, c, d, h, w))
    if red is None:
        L.check(L.load().xh_duse_gate_fwd(_stream(), _dt(x), _p(x), bs, _p(ch), _p(sp), _vol(sp)[5], _p(u), _vol(u)[5], n, c,
                                          d * h * w), "xh_duse_gate_fwd")
    else:
        L.check(L.load().xh_duse_gate_fwd_stats(_stream(), _dt(x), _p(x), bs, _p(ch), _p(sp), _vol(sp)[5], _p(u), _vol(u)[5], n, c,
                                                d * h * w, _p(red)), "xh_duse_gate_fwd_stats")
    return u


def duse_gate_bwd_fuses(c):
    """True when xh_duse_gate_bwd runs as ONE pass for c channels and can then also store dsp through the sigmoid's backward."""
    return bool(L.load().xh_duse_gate_bwd_fuses(int(c)))


def duse_gate_bwd(x, ch, sp, du, dsp_out, sigmoid_bwd=False):
    """sigmoid_bwd (only where duse_gate_bwd_fuses(C)): dsp_out = (sum_c du x) * sp (1 - sp) instead of the plain sum."""
    n, c, d, h, w, bs = _vol(x)
    dx = new_like(x, (n, c, d, h, w))
    dch = zeros_f64(x.device, (n, c))
    L.check(L.load().xh_duse_gate_bwd(_stream(), _dt(x), _p(x), bs, _p(ch), _p(sp), _vol(sp)[5], _p(du), _vol(du)[5], _p(dx),
                                      _vol(dx)[5], _p(dsp_out), _vol(dsp_out)[5], _p(dch), n, c, d * h * w, int(bool(sigmoid_bwd))),
            "xh_duse_gate_bwd")
    return dx, dch


FC_FOLD = [os.environ.get("XH_NO_FC_FOLD", "") == ""]      # A/B switch: DuSE's tiny dense layers inside the pair's gate / input-gradient passes


def duse_gate_fc(xpair, sp, red_in, p, red=None):
    """The pair's gate pass with the channel excitation derived in-kernel (xh_duse_gate_fc_fwd).  xpair (1, 2C, ...), sp (1, 2, ...),
    red_in (1, 2C, 2) raw sums; p: dict wc, bc, w1, b1, w2, b2.  Returns u (1, 2C, ...), ch (2, C), g (1, C), means (1, 2C)."""
    n, c2, d, h, w, bs = _vol(xpair)
    c = c2 // 2
    u = torch.empty_like(xpair, memory_format=torch.contiguous_format)
    ch = torch.empty((2, c), dtype=torch.float32, device=xpair.device)
    g = torch.empty((1, c), dtype=torch.float32, device=xpair.device)
    means = torch.empty((1, c2), dtype=torch.float32, device=xpair.device)
    dhw = d * h * w
    L.check(L.load().xh_duse_gate_fc_fwd(_stream(), _dt(xpair), _p(xpair), c * dhw, _p(sp), dhw, _p(u), c * dhw, c, dhw, _p(red_in),
                                         _p(p["wc"]), _p(p["bc"]), _p(p["w1"]), _p(p["b1"]), _p(p["w2"]), _p(p["b2"]), _p(ch), _p(g),
                                         _p(means), _p(red)), "xh_duse_gate_fc_fwd")
    return u, ch, g, means


def rank1_add_fc(dx, d1, w, means, g, ch, dch, p, grads):
    """dx (1, 2C, ...) += w[c] * d1 + d(mean)[c]; the pooled-mean gradient and the six dense-layer gradients come from the same launch
    (xh_rank1_add_fc).  p: dict wc, w1, w2; grads: dict wc, bc, w1, b1, w2, b2 (accumulated into)."""
    n, c2, d, h, ww, bs = _vol(dx)
    L.check(L.load().xh_rank1_add_fc(_stream(), _dt(dx), _p(dx), bs, _p(d1), _vol(d1)[5], _p(w), c2, d * h * ww, _p(means), _p(g), _p(ch),
                                     _p(dch), _p(p["wc"]), _p(p["w1"]), _p(p["w2"]), _p(grads["wc"]), _p(grads["bc"]), _p(grads["w1"]),
                                     _p(grads["b1"]), _p(grads["w2"]), _p(grads["b2"])), "xh_rank1_add_fc")
    return dx


def rank1_add(dx, d1, w, k):
    n, c, d, h, ww, bs = _vol(dx)
    L.check(L.load().xh_rank1_add(_stream(), _dt(dx), _p(dx), bs, _p(d1), _vol(d1)[5], _p(w), _p(k), n, c, d * h * ww), "xh_rank1_add")
    return dx


def compose_atten_fwd(params, ns, ne, e):
    """params: seg_w, seg_b, seg2_w, seg2_b, enc_w, enc_b, enc2_w, enc2_b (fp32, contiguous)."""
    seg_w = params[0]
    k3 = seg_w[0].numel()
    w = torch.empty((2, ne, k3), dtype=torch.float32, device=seg_w.device)
    b = torch.empty(2, dtype=torch.float32, device=seg_w.device)
    L.check(L.load().xh_compose_atten_fwd(_stream(), *[_p(t) for t in params], ns, ne, e, k3, _p(w), _p(b)), "xh_compose_atten_fwd")
    return w, b


def compose_atten_bwd(params, ns, ne, e, gw, gb, grads):
    seg_w, seg_b, seg2_w, _, enc_w, enc_b, enc2_w, _ = params
    k3 = seg_w[0].numel()
    L.check(L.load().xh_compose_atten_bwd(_stream(), _p(seg_w), _p(seg_b), _p(seg2_w), _p(enc_w), _p(enc_b), _p(enc2_w), ns, ne, e,
                                          k3, _p(gw), _p(gb), *[_p(g) for g in grads]), "xh_compose_atten_bwd")


def compose_multi(bwd, atten, duse, head, zero=None, sep=()):
    """All parameter compositions of a step in one launch (xh_compose_multi).
    atten: list of dicts(params=8 tensors, ns, ne, e, w, b[, grads=8 buffers, gw, gb]); duse: dicts(params=10, c, out=4[, grads=10,
    gout=4]); head: dict(wf, bf, ws, bs, w, b[, dwf, dbf, dws, dbs, gw, gb]) or None.  zero: an fp32 tensor the launch also clears."""
    """sep: dicts(dw (C,1,k,k,k), pw (C,C,1,1,1), w (C,C,k,k,k)[, gw, g_dw, g_pw]): depthwise o pointwise as one dense conv."""
    na, nd = len(atten), len(duse)
    first = atten[0]["params"][0] if na else duse[0]["params"][0] if nd else head["wf"] if head is not None else sep[0]["dw"]
    if not first.is_cuda:
        raise RuntimeError("xlstm_hved_amd ops need device tensors (the HIP library is the only compute path)")
    aj = (L.AttenJob * max(na, 1))()
    for j, a in zip(aj, atten):
        j.p = (C.c_void_p * 8)(*[_p(t) for t in a["params"]])
        j.NS, j.NE, j.E, j.K3 = a["ns"], a["ne"], a["e"], a["params"][0][0].numel()
        if bwd:
            j.g = (C.c_void_p * 8)(*[_p(t) for t in a["grads"]])
            j.gw, j.gb = _p(a["gw"]), _p(a["gb"])
        else:
            j.w, j.b = _p(a["w"]), _p(a["b"])
    dj = (L.DuseJob * max(nd, 1))()
    for j, d in zip(dj, duse):
        j.p = (C.c_void_p * 10)(*[_p(t) for t in d["params"]])
        j.C = d["c"]
        if bwd:
            j.g = (C.c_void_p * 10)(*[_p(t) for t in d["grads"]])
            j.gout = (C.c_void_p * 4)(*[_p(t) for t in d["gout"]])
        else:
            j.out = (C.c_void_p * 4)(*[_p(t) for t in d["out"]])
    hj = L.HeadJob()
    if head is not None:
        for k in ("wf", "bf", "ws", "bs") + (("dwf", "dbf", "dws", "dbs", "gw", "gb") if bwd else ("w", "b")):
            setattr(hj, k, _p(head[k]))
        hj.Co, hj.Cm, hj.Ci = head["wf"].shape[0], head["ws"].shape[0], head["ws"].shape[1]
    sj = (L.SepJob * max(len(sep), 1))()
    for j, q in zip(sj, sep):
        j.dw, j.pw, j.C, j.K3 = _p(_f32(q["dw"], "dw")), _p(_f32(q["pw"], "pw")), q["dw"].shape[0], q["dw"][0].numel()
        if bwd:
            j.gw, j.g_dw, j.g_pw = _p(q["gw"]), _p(q["g_dw"]), _p(q["g_pw"])
        else:
            j.w = _p(q["w"])
    L.check(L.load().xh_compose_multi(_stream(), int(bwd), na, C.cast(aj, C.c_void_p), nd, C.cast(dj, C.c_void_p),
                                      int(head is not None), C.cast(C.pointer(hj), C.c_void_p), len(sep), C.cast(sj, C.c_void_p),
                                      _p(zero), zero.numel() if zero is not None else 0), "xh_compose_multi")


def _ptr10(ts):
    return (C.c_void_p * 10)(*[_p(t) for t in ts])


def compose_duse_fwd(params, c):
    dev = params[0].device
    sqw = torch.empty(2 * c, dtype=torch.float32, device=dev)
    sqb = torch.empty(1, dtype=torch.float32, device=dev)
    adjw = torch.empty((2, 1, 3, 3, 3), dtype=torch.float32, device=dev)
    adjb = torch.empty(2, dtype=torch.float32, device=dev)
    L.check(L.load().xh_compose_duse_fwd(_stream(), C.byref(_ptr10(params)), c, _p(sqw), _p(sqb), _p(adjw), _p(adjb)), "xh_compose_duse_fwd")
    return sqw.view(1, 2 * c, 1, 1, 1), sqb, adjw, adjb


def compose_duse_bwd(params, c, dsqw, dsqb, dadjw, dadjb, grads):
    L.check(L.load().xh_compose_duse_bwd(_stream(), C.byref(_ptr10(params)), c, _p(dsqw), _p(dsqb), _p(dadjw), _p(dadjb),
                                         C.byref(_ptr10(grads))), "xh_compose_duse_bwd")


def duse_fc_fwd(red_r, red_s, count, n, c, p, ch_out=None):
    """-> g, ch1, ch2 (n, c) and the pooled means (n, 2c) for duse_fc_bwd (own storage: red_r / red_s may be scratch).
    ch_out: a (2n, c) fp32 tensor whose halves receive ch1 / ch2 (the recon | seg pair gated in one launch)."""
    g = torch.empty((n, c), dtype=torch.float32, device=red_r.device)
    if ch_out is not None:
        ch1, ch2 = ch_out[:n], ch_out[n:]
    else:
        ch1, ch2 = (torch.empty((n, c), dtype=torch.float32, device=red_r.device) for _ in range(2))
    means = torch.empty((n, 2 * c), dtype=torch.float32, device=red_r.device)
    L.check(L.load().xh_duse_fc_fwd(_stream(), _p(red_r), _p(red_s), count, n, c, _p(p["wc"]), _p(p["bc"]), _p(p["w1"]), _p(p["b1"]),
                                    _p(p["w2"]), _p(p["b2"]), _p(g), _p(ch1), _p(ch2), _p(means)), "xh_duse_fc_fwd")
    return g, ch1, ch2, means


def duse_fc_bwd(means, count, n, c, p, g, ch1, ch2, dch1, dch2, grads, dm_out=None):
    """means: from duse_fc_fwd.  grads: dict of fp32 buffers (wc, bc, w1, b1, w2, b2) the kernel ACCUMULATES into."""
    if dm_out is not None:                    # (n == 1: the halves of one (1, 2c) row)
        dmr, dms = dm_out[:, :c], dm_out[:, c:]
        if n != 1:
            raise ValueError("duse_fc_bwd: dm_out needs n == 1")
    else:
        dmr, dms = (torch.empty((n, c), dtype=torch.float32, device=g.device) for _ in range(2))
    L.check(L.load().xh_duse_fc_bwd(_stream(), None, None, count, n, c, _p(p["wc"]), _p(p["w1"]), _p(p["w2"]), _p(g), _p(ch1),
                                    _p(ch2), _p(dch1), _p(dch2), _p(grads["wc"]), _p(grads["bc"]), _p(grads["w1"]), _p(grads["b1"]),
                                    _p(grads["w2"]), _p(grads["b2"]), _p(dmr), _p(dms), _p(means)), "xh_duse_fc_bwd")
    return dmr, dms


def skr_tail(t, x, sc, sh, w2):
    n, c, d, h, w, _ = _vol(x)
    a = new_like(x, (n, 1, d, h, w))
    L.check(L.load().xh_skr_tail_fwd(_stream(), _dt(x), _p(t), _p(x), _p(sc), _p(sh), _p(w2), _p(a), n, c, d * h * w), "xh_skr_tail_fwd")
    return a


BN_FOLD = [os.environ.get("XH_NO_BN_FOLD", "") == ""]      # A/B switch: the skip-return ResBlock's BatchNorm finalisations inside their consumers


def skr_tail_bn(t, x, red, gamma, beta, running_mean, running_var, steps, w2):
    """skr_tail with the training-mode BatchNorm in front of it finalised in the same launch (one sample, xh_skr_tail_bn_fwd);
    returns (a, sc, sh, mean, rstd)."""
    n, c, d, h, w, _ = _vol(x)
    if n != 1:
        raise ValueError("skr_tail_bn: one sample per launch")
    a = new_like(x, (n, 1, d, h, w))
    sc, sh, mean, rstd = (torch.empty((1, c), dtype=torch.float32, device=x.device) for _ in range(4))
    L.check(L.load().xh_skr_tail_bn_fwd(_stream(), _dt(x), _p(t), _p(x), _p(red), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                        int(steps), _p(w2), _p(a), c, d * h * w, _p(sc), _p(sh), _p(mean), _p(rstd)), "xh_skr_tail_bn_fwd")
    return a, sc, sh, mean, rstd


def skr_tail_bwd(t, x, sc, sh, w2, a, da, dw2_out=None, dx_acc=None):
    """dw2_out: contiguous fp32 (2,) buffer the two weight gradients are ACCUMULATED into (the parameter's gradient); else a
    fresh fp64 pair is returned.  dx_acc: a gradient buffer of x's shape the residual-branch gradient is ADDED into (returned
    as dx); else a new tensor."""
    n, c, d, h, w, _ = _vol(x)
    if dx_acc is not None and (dx_acc.shape != x.shape or dx_acc.dtype != x.dtype or not dx_acc.is_contiguous()):
        raise ValueError("skr_tail_bwd: dx_acc must be a contiguous tensor of x's shape and type")
    dtg, dx = torch.empty_like(t), (dx_acc if dx_acc is not None else torch.empty_like(x))
    dw2 = None if dw2_out is not None else zeros_f64(x.device, (2,))
    if dw2_out is not None and (dw2_out.dtype != torch.float32 or dw2_out.numel() != 2 or not dw2_out.is_contiguous()):
        raise ValueError("skr_tail_bwd: dw2_out must be a contiguous fp32 tensor of 2 elements")
    L.check(L.load().xh_skr_tail_bwd(_stream(), _dt(x), _p(t), _p(x), _p(sc), _p(sh), _p(w2), _p(a), _p(da), _p(dtg), _p(dx), _p(dw2),
                                     n, c, d * h * w, int(dx_acc is not None), _p(dw2_out)), "xh_skr_tail_bwd")
    return dtg, dx, dw2


# ----------------------------------------------------------------------------------------------- ViL
def _vil_struct(tensors):
    s = L.VilParams()
    for name in L.VIL_FIELDS:
        setattr(s, name, _p(_f32(tensors[name], name)))
    return s


def vil_fwd(xa, xb, params, add_xa=True, nh=4):
    """out = (xa if add_xa) + ViLBlock(xa + xb); xa/xb (B,C,D,H,W).  Returns (out, workspace)."""
    n, c, d, h, w, _ = _vol(xa)
    xa = xa.contiguous()
    xb = xb.contiguous() if xb is not None else None
    s = d * h * w
    lib = L.load()
    ws = torch.empty(lib.xh_vil_workspace_floats(n, s, c), dtype=torch.float32, device=xa.device)
    out = torch.empty_like(xa)
    ps = _vil_struct(params)
    L.check(lib.xh_vil_fwd(_stream(), _dt(xa), _p(xa), _p(xb), _p(out), n, s, c, nh, int(add_xa), C.byref(ps), _p(ws)), "xh_vil_fwd")
    return out, ws


def vil_bwd(xa, xb, dout, params, ws, grads, nh=4):
    """grads: dict of fp32 buffers (L.VIL_FIELDS) the kernels ACCUMULATE into."""
    n, c, d, h, w, _ = _vol(xa)
    s = d * h * w
    dout = dout.contiguous()
    dxin = torch.empty_like(dout)
    ps, gs = _vil_struct(params), _vil_struct(grads)
    L.check(L.load().xh_vil_bwd(_stream(), _dt(dout), _p(xa), _p(xb), _p(dout), _p(dxin), n, s, c, nh, C.byref(ps), C.byref(gs), _p(ws)),
            "xh_vil_bwd")
    return dxin


# ----------------------------------------------------------------------------------------------- loss / metric epilogues
def _dt_of(t):
    return _dt(t)


def pair_sums(a, b=None, bval=0.0, thr=None, red=None):
    """(N, C, 6) fp64 sums over DHW of (a'b, a'^2, b^2, (a'-b)^2, a', b); a' = (a > thr) when thr is given.  b: a tensor of
    a's dtype or fp32, or None (= the constant bval).  `red`: zeroed (N, C, 6) fp64 destination (default: a slice of the
    per-forward scratch arena -- the sums are consumed by the finalisation launch that follows)."""
    n, c, d, h, w, a_bs = _vol(a)
    if red is None:
        red = zeros_f64(a.device, (n, c, 6))
    b_dt, b_bs = _dt(a), 0
    if b is not None:
        if tuple(b.shape) != tuple(a.shape):
            raise ValueError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        b_dt, b_bs = _dt(b), _vol(b)[5]
    L.check(L.load().xh_pair_sums(_stream(), _dt(a), _p(a), a_bs, b_dt, _p(b), b_bs, float(bval), n, c, d * h * w,
                                  int(thr is not None), float(thr or 0.0), _p(red)), "xh_pair_sums")
    return red


def lincomb(a, b, ca, cb, cc=None, bval=0.0, gscale=None):
    """out = gscale * (ca[n,c]*a + cb[n,c]*b + cc[n,c]) in a's dtype (b: tensor of a's dtype / fp32, or None = bval;
    gscale: optional fp32 device scalar)."""
    n, c, d, h, w, a_bs = _vol(a)
    out = torch.empty_like(a, memory_format=torch.contiguous_format)
    b_dt, b_bs = _dt(a), 0
    if b is not None:
        b_dt, b_bs = _dt(b), _vol(b)[5]
    L.check(L.load().xh_lincomb(_stream(), _dt(a), _p(a), a_bs, b_dt, _p(b), b_bs, float(bval), _p(out), _vol(out)[5], n, c,
                                d * h * w, _p(_f32(ca, "ca")), _p(_f32(cb, "cb")), _p(cc), _p(gscale), 0), "xh_lincomb")
    return out


def loss_finalize(kind, red, count=1.0, eps=1e-6):
    """kind 0 DiceLoss / 1 mean squared difference: (loss[1], ca[N,C], cb[N,C]); kind 2 thresholded Dice: metric[C]."""
    n, c = red.shape[:2]
    out = torch.empty(c if kind == 2 else 1, dtype=torch.float32, device=red.device)
    ca = cb = None
    if kind == 3:
        L.check(L.load().xh_loss_finalize(_stream(), 3, _p(red), n, c, float(count), float(eps), _p(out), None, None), "xh_loss_finalize")
        return out
    if kind == 4:                        # eps carries the weights tensor here
        L.check(L.load().xh_loss_finalize(_stream(), 4, _p(red), n, c, 1.0, 0.0, _p(out), _p(_f32(count, "weights")), None), "xh_loss_finalize")
        return out
    if kind < 2:
        ca, cb = (torch.empty((n, c), dtype=torch.float32, device=red.device) for _ in range(2))
    L.check(L.load().xh_loss_finalize(_stream(), kind, _p(red), n, c, float(count), float(eps), _p(out), _p(ca), _p(cb)),
            "xh_loss_finalize")
    return (out, ca, cb) if kind < 2 else out


def scalar_lincomb(terms, coefs):
    """sum_i coefs[i] * terms[i] of <= 16 device scalars (fp32 or fp64 tensors of one element) in ONE launch: fp32 (1,)."""
    n = len(terms)
    for t in terms:
        if t.numel() != 1 or t.dtype not in (torch.float32, torch.float64):
            raise TypeError("scalar_lincomb takes one-element fp32 / fp64 tensors")
    out = torch.empty(1, dtype=torch.float32, device=terms[0].device)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in terms])
    f64 = (C.c_int * n)(*[int(t.dtype == torch.float64) for t in terms])
    cs = (C.c_double * n)(*[float(c) for c in coefs])
    L.check(L.load().xh_scalar_lincomb(_stream(), n, ptrs, f64, cs, _p(out)), "xh_scalar_lincomb")
    return out


def scalar_fanout(coefs, g):
    """(fp32 (n,), fp64 (n,)) = coefs * g[0]: the gradients of scalar_lincomb's terms in one launch (g: fp32 device scalar)."""
    n = len(coefs)
    o32 = torch.empty(n, dtype=torch.float32, device=g.device)
    o64 = torch.empty(n, dtype=torch.float64, device=g.device)
    cs = (C.c_double * n)(*[float(c) for c in coefs])
    L.check(L.load().xh_scalar_fanout(_stream(), n, cs, _p(_f32(g, "g")), _p(o32), _p(o64)), "xh_scalar_fanout")
    return o32, o64


def kld_fwd(mu, lv, keep, red=None):
    n, five, L_, d, h, w = mu.shape
    if red is None:
        red = torch.zeros(1, dtype=torch.float64, device=mu.device)
    L.check(L.load().xh_kld_fwd(_stream(), _dt(mu), _p(mu), _p(lv), _p(keep), n, L_, d * h * w, _p(red)), "xh_kld_fwd")
    return red


def kld_bwd(mu, lv, keep, scale, gscale=None):
    n, five, L_, d, h, w = mu.shape
    dmu, dlv = torch.empty_like(mu), torch.empty_like(lv)
    L.check(L.load().xh_kld_bwd(_stream(), _dt(mu), _p(mu), _p(lv), _p(keep), n, L_, d * h * w, float(scale), _p(gscale), _p(dmu),
                                _p(dlv)), "xh_kld_bwd")
    return dmu, dlv


def nested_weight(seg):
    n, c, d, h, w, bs = _vol(seg)
    if c != 3:
        raise ValueError("nested weights need the 3 region channels (WT, TC, ET)")
    out = new_like(seg, (n, 1, d, h, w))
    L.check(L.load().xh_nested_weight(_stream(), _dt(seg), _p(seg), bs, _p(out), _vol(out)[5], n, d * h * w), "xh_nested_weight")
    return out


def multi_sum(tensors, red, row0, rows):
    """One launch: red[(row0[t] + b % rows[t]), 0, 4] += partial sums of tensors[t] (all of one storage type, <= 16)."""
    n = len(tensors)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nums = (C.c_longlong * n)(*[t.numel() for t in tensors])
    r0 = (C.c_int * n)(*row0)
    rs = (C.c_int * n)(*rows)
    L.check(L.load().xh_multi_sum(_stream(), _dt(tensors[0]), n, ptrs, nums, r0, rs, _p(red)), "xh_multi_sum")


def multi_fill(metas, values, gscale=None):
    """One launch: new tensors of the (shape, dtype, device) metas (one dtype, <= 16), tensor t filled with values[t] * gscale."""
    outs = [torch.empty(shape, dtype=dtype, device=device) for shape, dtype, device in metas]
    n = len(outs)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    nums = (C.c_longlong * n)(*[t.numel() for t in outs])
    vals = (C.c_float * n)(*[float(v) for v in values])
    L.check(L.load().xh_multi_fill(_stream(), _dt(outs[0]), n, ptrs, nums, vals, _p(gscale)), "xh_multi_fill")
    return outs


def fill(shape, value, dtype, device, gscale=None):
    out = torch.empty(shape, dtype=dtype, device=device)
    L.check(L.load().xh_fill(_stream(), _dt(out), _p(out), out.numel(), float(value), _p(gscale)), "xh_fill")
    return out
