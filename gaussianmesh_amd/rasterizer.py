"""Operator API of the differentiable Gaussian rasterizer on PyTorch-ROCm tensors.

Mirrors gaussian_renderer/diff_gaussian_rasterizater/__init__.py of the reference:
  GaussianRasterizationSettings (:6-18), _RasterizeGaussians fwd/bwd (:23-124), GaussianRasterizer (:126-174),
  NewGaussianRasterizer (:280-328; same op with the SH stride M forced to 16,
  rasterize_points_deformed.py:157,230).
Same names, argument order, shapes, dtypes and error behaviour; tensors are torch tensors on a HIP device and
the work is done by libgmesh_hip.so through its C ABI (include/gmesh_hip.h).  Jittor's nn.Module.execute is
provided as an alias of forward.
"""
import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    work_hint: Optional[torch.Tensor] = None     # extension (new_work_hint()): per-tile cost memory of this camera / view stream



def _ptr(t):
    return None if t is None else t.data_ptr()


def _prep(t, device):
    """contiguous float32 on `device`, or None for an absent / empty optional input."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise ValueError("rasterizer input on %s but means3D on %s" % (t.device, device))
    if t.dtype is torch.float32 and t.is_contiguous():      # the usual case: nothing to convert (only the pointer is used; a render
        return t                                            # loop issues a dozen of these per frame, each conversion a dispatcher trip)
    return t.detach().contiguous().float()


class _on:
    """`with torch.cuda.device(d)` for the usual case that d already is the current device: one C call instead of the context
    manager's four (a pipelined render loop enters it three times per frame)."""
    __slots__ = ("guard",)

    def __init__(self, device):
        self.guard = None if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *a):
        if self.guard is not None:
            return self.guard.__exit__(*a)


def _stream(device):
    return torch._C._cuda_getCurrentRawStream(device.index)


_STREAM_OBJECTS = {}


def _current_stream(device):
    """torch.cuda.current_stream(device), with the Stream wrapper cached per raw handle (the torch call spends ~10 us in
    device-index bookkeeping; a render loop asks three times per frame)."""
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    st = _STREAM_OBJECTS.get(key)
    if st is None:
        st = torch.cuda.current_stream(device)
        if len(_STREAM_OBJECTS) < 256:
            _STREAM_OBJECTS[key] = st
    return st


# Instance emission policy (include/gmesh_hip.h): an argument of every call below (`emission_policy=`); None takes this
# module-level default.  The C library keeps no policy state.  The built-in default is "auto": lists per 32-px parent tile
# (policy 2) while the image has at most 2048 of them - the tile sort is then ONE 11-bit pass - and per 64-px parent (policy 3)
# above that (measured on the C3 cloud, frames/s with policy 2 / 3: 1080p 4460 / 4330, 3840x2160 2110 / 2250, C5 training
# iteration 4.78 / 4.66 ms; 960x540 with policy 1 / 2: 5290 / 5520).  GM_EMISSION_MODE=0..3 fixes it for A/B runs.
def _parse_policy(mode):
    if isinstance(mode, str) and mode.strip().lower() == "auto":
        return "auto"
    return min(3, max(0, int(mode)))


_default_policy = [_parse_policy(os.environ.get("GM_EMISSION_MODE", "auto"))]


def auto_emission_policy(width, height):
    """Policy 2 when the image has at most 2048 parent tiles of 32 px, otherwise 3."""
    gx, gy = (int(width) + 15) // 16, (int(height) + 15) // 16
    return 2 if ((gx + 1) // 2) * ((gy + 1) // 2) <= 2048 else 3


def set_default_emission_policy(mode):
    """mode: 0 (the reference's emission), 1, 2, 3, or "auto" (by image size, see above)."""
    _default_policy[0] = _parse_policy(mode)


def get_default_emission_policy(width=None, height=None):
    """The default policy; with an image size an "auto" default is resolved for it."""
    d = _default_policy[0]
    if d == "auto" and width is not None and height is not None:
        return auto_emission_policy(width, height)
    return d


def _pol(p, width, height):
    if p is None:
        p = _default_policy[0]
    return auto_emission_policy(width, height) if p == "auto" else int(p)


class RasterWorkspace:
    """Caller-owned scratch (geometry / image / binning byte buffers) that is reused across forward calls and
    grows geometrically, so a render loop performs no device allocation in steady state (the reference allocates
    and zero-fills all three buffers on every call, rasterize_points.py:118-121, 192-194).  Only valid while no
    backward pass still needs the buffers: the autograd operator uses a fresh set whenever a gradient is required.

    ONE forward at a time may use a workspace: *_begin() marks it in flight and PendingForward.finish() releases it;
    beginning another frame on a workspace whose previous frame is unfinished raises (the new frame would overwrite the
    geometry the old one still has to render from).  A pipelined loop keeps one workspace per frame in flight."""

    def __init__(self, growth=1.25):
        self.growth = growth
        self._bufs = {}
        self._pinned = None
        self._status = None
        self._event = None
        self.in_flight = None
        self.capacity = 0            # instances the binning buffer holds (sync-free mode)

    def pinned_counter(self):
        """page-locked int32[1] receiving the instance count of an asynchronous forward"""
        if self._pinned is None:
            self._pinned = torch.zeros((1,), dtype=torch.int32).pin_memory()
        return self._pinned

    def pinned_status(self):
        if self._status is None:
            self._status = torch.zeros((4,), dtype=torch.int32).pin_memory()
        return self._status

    def status_event(self):
        """the event recorded behind a sync-free frame's status words: one per workspace (a workspace carries one frame at a
        time, and its event is only recorded again after that frame has been checked and released it)"""
        if self._event is None:
            self._event = torch.cuda.Event()
        return self._event

    def get(self, name, nbytes, device):
        b = self._bufs.get(name)
        if b is None or b.device != device or b.numel() < nbytes:
            b = torch.empty((int(nbytes * self.growth) + 256,), dtype=torch.uint8, device=device)
            self._bufs[name] = b
        return b

    def acquire(self, owner):
        if self.in_flight is not None:
            raise _lib.GmeshError("RasterWorkspace: the previous frame begun on this workspace has not been finished; use one "
                                  "workspace per frame in flight")
        self.in_flight = owner

    def release(self, owner):
        if self.in_flight is owner:
            self.in_flight = None


_STREAMS_SEEN = set()
_QUEUE_WARNED = [False]


def _note_stream(stream):
    """Frames pipelined over three or more HIP streams only overlap if the runtime has a hardware queue for each: HIP multiplexes
    streams onto FOUR queues by default and streams that share one serialise (four streams: 4160 frames/s on 4 queues, 5010 on 8,
    DESIGN.md section 5).  gaussianmesh_amd.configure_runtime() exports GPU_MAX_HW_QUEUES=8 when the integrator calls it before the
    first HIP call (the package itself never touches the environment).  Warn, once, when the third stream shows up and the variable is
    not set (or was set too late to be read)."""
    if _QUEUE_WARNED[0]:
        return
    _STREAMS_SEEN.add(stream.cuda_stream)
    if len(_STREAMS_SEEN) >= 3:
        _QUEUE_WARNED[0] = True
        import os
        import warnings
        import gaussianmesh_amd as _pkg
        cfg = _pkg.RUNTIME_CONFIG
        if "GPU_MAX_HW_QUEUES" not in os.environ or (cfg.get("hw_queues") and not cfg.get("applied") and "AFTER" in cfg.get("note", "")):
            warnings.warn("gaussianmesh_amd: frames are being issued on %d HIP streams, but GPU_MAX_HW_QUEUES was not set before the HIP "
                          "runtime started; the runtime multiplexes streams onto 4 hardware queues by default and streams sharing a queue "
                          "serialise (-17 %% on the four-stream loop). Call gaussianmesh_amd.configure_runtime() (or export "
                          "GPU_MAX_HW_QUEUES=8) before the first torch.cuda call." % len(_STREAMS_SEEN), RuntimeWarning, stacklevel=3)





def new_work_hint(width, height, device):
    """Zeroed work-hint buffer for PendingForward.finish(work_hint=...) / gm_forward_1_geom: one per view stream."""
    return torch.zeros((_lib.lib().gm_work_hint_bytes(width, height) // 4,), dtype=torch.int32, device=device)


class DepthPlan:
    """Depth-bucket tables of ONE view stream (consecutive cameras of an orbit / an edit session), handed to
    forward_deformed_begin(depth_plan=...): from the stream's second frame on the fused pass places every visible Gaussian in
    its depth bucket itself (gm_forward_0_deformed_stream_async, direct placement) instead of running the depth partition.
    Never changes an image: a frame the direct placement refuses (check() -> not fitted) is begun again on the partition path
    by finish().  `refused` counts those frames."""

    def __init__(self, device):
        self.buf = torch.zeros((_lib.lib().gm_depth_plan_bytes() // 4,), dtype=torch.int32, device=device)
        self.primed = False
        self.refused = 0


def new_depth_plan(device):
    return DepthPlan(device)


class PendingForward:
    """Handle returned by rasterize_forward_begin() / forward_deformed_begin(): everything up to the instance count is
    enqueued.  finish() waits for the count (one 4-byte pinned-memory read-back, normally long complete), sizes the
    binning buffer and enqueues emission, tile sort and blend.  finish(sync_free=True) does not wait at all: the binning
    buffer of the workspace is used at its current capacity and the kernels read the count on the device; check()
    later tells whether the frame fitted (False: call finish() again - it then takes the exact path)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.status_event = None
        self.checked = None              # (fitted, num_rendered) once check() has consumed the status
        self.result = None
        self.known_count = None
        self.image_only = False
        self.exact_exponent = False
        self.work_hint = None
        self.direct = False              # begun with direct depth placement (DepthPlan); rebegin re-issues the first half on the partition path
        self.rebegin = None
        self.refusal = 0                 # status word 3 of the checked frame: 1 capacity / policy, 2 direct placement

    def _geom(self, binning, num_rendered, capacity, status=None):
        lib = _lib.lib()
        a = self.args
        _lib.check(lib.gm_forward_1_geom(self.policy, _ptr(self.geom), _ptr(binning), _ptr(self.img), a["P"], num_rendered, capacity,
                                         _ptr(a["bg"]), a["W"], a["H"], _ptr(self.color), a["debug"], self.stream.cuda_stream,
                                         None if status is None else status.data_ptr(),
                                         (1 if self.image_only else 0) | (2 if self.exact_exponent else 0),
                                         None if self.work_hint is None else self.work_hint.data_ptr()))

    def finish(self, sync_free=False, capacity=0, image_only=False, work_hint=None, exact_exponent=False):
        """See _finish().  A failure anywhere in here (allocation, a C-side error) releases the workspace: one failed frame must
        not leave it marked in flight for ever.
        exact_exponent (GM_FWD_EXACT_EXPONENT): the forward of a training step - the blend evaluates its exponents with the
        backward pass's per-pixel expression, so that both halves take the same alpha >= 1/255 decisions (the autograd operator
        sets it whenever a gradient is required)."""
        if exact_exponent and image_only:
            raise _lib.GmeshError("exact_exponent is for a forward that a backward pass follows: not together with image_only")
        self.exact_exponent = self.exact_exponent or bool(exact_exponent)      # (a repeated finish() of a refused frame keeps it)
        try:
            return self._finish(sync_free, capacity, image_only, work_hint)
        except Exception:
            if self.workspace is not None:
                self.workspace.release(self)
            raise

    def _finish(self, sync_free=False, capacity=0, image_only=False, work_hint=None):
        """capacity (sync_free without a workspace): instances the freshly allocated binning buffer shall hold.
        image_only (GM_FWD_IMAGE_ONLY): a frame no backward pass follows - the blend writes the colour image and leaves the
        per-pixel final transmittance / contributor count of the image state alone; the returned img must not be handed
        to rasterize_backward.
        work_hint (new_work_hint()): per-tile cost memory shared by the consecutive frames of one view stream; the blend's
        dispatch order then follows what tiles cost in recent frames.  Never changes an image."""
        self.image_only = bool(image_only)
        if work_hint is not None:
            key = (work_hint.data_ptr(), work_hint.numel(), self.args["W"], self.args["H"], self.args["device"])
            if key not in _CHECKED_HINTS:                     # (a render loop hands the same buffer over every frame)
                need = _lib.lib().gm_work_hint_bytes(self.args["W"], self.args["H"])
                if work_hint.dtype != torch.int32 or not work_hint.is_contiguous() or work_hint.numel() * 4 < need or work_hint.device != self.args["device"]:
                    raise _lib.GmeshError("work_hint: contiguous int32 tensor of gm_work_hint_bytes(W, H) bytes on the frame's device expected")
                if len(_CHECKED_HINTS) > 64:
                    _CHECKED_HINTS.clear()
                _CHECKED_HINTS.add(key)
        self.work_hint = work_hint
        lib = _lib.lib()
        a = self.args
        device = a["device"]
        ws = self.workspace
        if sync_free and ws is not None and ws.capacity > 0 and a["P"] > 0 and self.status_event is None and self.checked is None:
            nbytes = lib.gm_binning_bytes(ws.capacity)
            binning = ws._bufs.get("binning")
            if binning is not None and binning.device == device and binning.numel() >= nbytes:
                # steady state of a pipelined render loop: the buffer exists, nothing is allocated, so nothing here depends on
                # torch's current stream - the launches take the frame's stream explicitly
                with _on(device):
                    self._geom(binning, -1, ws.capacity, ws.pinned_status())
                    self.status_event = ws.status_event()
                    self.status_event.record(self.stream)
                self.binning = binning
                self.result = (-1, self.color, self.radii, self.geom, binning, self.img)
                return self.result
        if self.direct and self.refusal == 2 and self.rebegin is not None:
            # the direct depth placement refused the frame (no table yet / stale table / piles of equal depths): the first half again,
            # on the partition path, into the same buffers; the exact path below completes it
            self.rebegin()
            self.direct, self.refusal, self.known_count, self.count_host = False, 0, None, None
        with _on(device), torch.cuda.stream(self.stream):
            if sync_free and ws is None and capacity > 0 and a["P"] > 0 and self.status_event is None and self.checked is None:
                binning = torch.empty((lib.gm_binning_bytes(capacity),), dtype=torch.uint8, device=device)
                self.status_host = _PINNED_STATUS.pop() if _PINNED_STATUS else torch.zeros((4,), dtype=torch.int32).pin_memory()
                self._geom(binning, -1, capacity, self.status_host)      # the blend kernel writes the status words itself
                self.status_event = torch.cuda.Event()
                self.status_event.record(self.stream)
                self.capacity = capacity
                self.result = (-1, self.color, self.radii, self.geom, binning, self.img)
                return self.result
            if sync_free and ws is not None and ws.capacity > 0 and a["P"] > 0 and self.status_event is None and self.checked is None:
                binning = ws.get("binning", lib.gm_binning_bytes(ws.capacity), device)
                self._geom(binning, -1, ws.capacity, ws.pinned_status())
                self.status_event = torch.cuda.Event()
                self.status_event.record(self.stream)
                self.binning = binning
                self.result = (-1, self.color, self.radii, self.geom, binning, self.img)
                return self.result
            if self.count_host is not None:
                self.event.synchronize()
                num_rendered = int(self.count_host[0])
            elif self.known_count is not None:                       # begun without the count copy; check() has read the status
                num_rendered = self.known_count
            else:                                                    # begun without the count copy and never checked: fetch it now
                st = torch.zeros((4,), dtype=torch.int32).pin_memory()
                _lib.check(lib.gm_forward_status_async(_ptr(self.geom), a["P"], st.data_ptr(), self.stream.cuda_stream))
                self.stream.synchronize()
                num_rendered = int(st[0])
            if ws is not None:
                ws.capacity = max(ws.capacity, int(num_rendered * ws.growth) + 1024)
                binning = ws.get("binning", lib.gm_binning_bytes(ws.capacity), device)
            else:
                binning = torch.empty((lib.gm_binning_bytes(num_rendered),), dtype=torch.uint8, device=device)
                if self.count_host is not None and len(_PINNED_POOL) < 64:
                    _PINNED_POOL.append(self.count_host)
            self._geom(binning, num_rendered, 0)
            if self.direct and self.rebegin is not None:
                # a direct-placement frame completed without the sync-free status protocol: look at its status here (one more
                # host wait on a path that waits for the count anyway) and take the partition path if it was refused
                st = torch.zeros((4,), dtype=torch.int32).pin_memory()
                _lib.check(lib.gm_forward_status_async(_ptr(self.geom), a["P"], st.data_ptr(), self.stream.cuda_stream))
                self.stream.synchronize()
                if int(st[3]) == 2:
                    self.rebegin()
                    _lib.check(lib.gm_forward_status_async(_ptr(self.geom), a["P"], st.data_ptr(), self.stream.cuda_stream))
                    self.stream.synchronize()
                    num_rendered = int(st[0])
                    self._geom(binning, num_rendered, 0)          # (the instance total does not depend on the depth path)
                self.direct = False
            self.status_event = None
            if a.get("prefiltered") and a["P"] > 0:         # the reference traps the kernel (auxiliary.h:155-159); here: an error
                st = torch.zeros((4,), dtype=torch.int32).pin_memory()
                _lib.check(lib.gm_forward_status_async(_ptr(self.geom), a["P"], st.data_ptr(), self.stream.cuda_stream))
                self.stream.synchronize()
                if int(st[1]):
                    raise _lib.GmeshError(_PREFILTER_MESSAGE)
        if ws is not None:
            ws.release(self)
        self.result = (num_rendered, self.color, self.radii, self.geom, binning, self.img)
        return self.result

    def check(self):
        """After finish(sync_free=True): wait for the frame's status and return (fitted, num_rendered).  When the instance
        count exceeded the workspace's capacity the image is the background; finish() renders the frame again, exactly.
        The status is consumed by the first call: a workspace carries ONE event and ONE pinned status block for the frames that
        pass through it, so a second look through an old handle must not read what a later frame left there - it returns the
        answer the first call recorded."""
        if self.status_event is None:
            if self.checked is not None:
                return self.checked
            return True, (self.result[0] if self.result else 0)
        ws = self.workspace
        if ws is not None and ws.in_flight is not self:
            raise _lib.GmeshError("PendingForward.check(): the workspace of this frame has been released and reacquired; its status "
                                  "words now belong to a later frame")
        self.status_event.synchronize()
        self.status_event = None
        if ws is None:
            st = self.status_host
            nr, refused, violated = int(st[0]), int(st[3]), int(st[1])
            self.refusal = refused
            self.known_count = nr
            if len(_PINNED_STATUS) < 64:
                _PINNED_STATUS.append(st)
            self.status_host = None
            self.checked = ((not refused), nr)
            if violated and self.args.get("prefiltered"):
                raise _lib.GmeshError(_PREFILTER_MESSAGE)
            return self.checked
        st = ws.pinned_status()
        nr, refused = int(st[0]), int(st[3])
        self.refusal = refused
        self.known_count = nr
        self.checked = ((not refused), nr)
        if int(st[1]) and self.args.get("prefiltered"):
            ws.release(self)
            raise _lib.GmeshError(_PREFILTER_MESSAGE)
        if refused:                      # still holds the workspace: finish() renders the frame again on the exact path
            return self.checked
        ws.release(self)
        return self.checked


_PREFILTER_MESSAGE = "Point is filtered although prefiltered is set. This shouldn't happen!"      # auxiliary.h:157
_CHECKED_HINTS = set()     # (pointer, size, W, H, device) of work-hint buffers already validated
_PINNED_POOL = []          # page-locked int32[1] counters of workspace-less forwards (allocating one per call costs ~0.1 ms)
_PINNED_STATUS = []        # page-locked int32[4] status words of sync-free forwards without a workspace


def _scratch(workspace, P, W, H, device):
    lib = _lib.lib()
    if workspace is not None:
        return (workspace.get("geom", lib.gm_geom_bytes(P), device), workspace.get("img", lib.gm_image_bytes(W, H), device),
                workspace.pinned_counter())
    count = _PINNED_POOL.pop() if _PINNED_POOL else torch.zeros((1,), dtype=torch.int32).pin_memory()
    return (torch.empty((lib.gm_geom_bytes(P),), dtype=torch.uint8, device=device),
            torch.empty((lib.gm_image_bytes(W, H),), dtype=torch.uint8, device=device), count)


def _count_event(stream):
    """an event the library re-records right behind the instance-count copy (needs a live hipEvent_t handle)"""
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def rasterize_forward_begin(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                            projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered=False,
                            debug=False, workspace=None, emission_policy=None, force_M=None):
    """First half of a forward without host synchronisation (gm_forward_0_async); returns a PendingForward.
    Typical loop: h_next = begin(frame i+1, workspace=ws[(i+1) % 2]); outputs = h_cur.finish()."""
    lib = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise _lib.GmeshError("gaussianmesh_amd rasterizer needs tensors on a HIP (cuda) device; there is no CPU path")
    policy = _pol(emission_policy, image_width, image_height)
    means3D = _prep(means3D, device)
    P = 0 if means3D is None else means3D.shape[0]
    sh, colors, scales, rotations, cov3D_precomp = (_prep(t, device) for t in (sh, colors, scales, rotations, cov3D_precomp))
    opacity = _prep(opacity, device)
    bg, viewmatrix, projmatrix, campos = (_prep(t, device) for t in (bg, viewmatrix, projmatrix, campos))
    H, W = int(image_height), int(image_width)
    M = 0
    if sh is not None:
        M = sh.shape[1] if sh.dim() == 3 else sh.numel() // (3 * max(P, 1))
    if force_M is not None and sh is not None and M != force_M:
        raise ValueError("NewGaussianRasterizer expects shs of shape [P,%d,3]" % force_M)
    stream = _current_stream(device)
    _note_stream(stream)
    h = PendingForward(policy=policy, workspace=workspace, stream=stream)
    if workspace is not None:
        workspace.acquire(h)
    try:
        with _on(device):
            color = torch.empty((3, H, W), dtype=torch.float32, device=device)
            radii = torch.empty((P,), dtype=torch.int32, device=device)
            geom, img, count_host = _scratch(workspace, P, W, H, device)
            event = _count_event(stream)
            _lib.check(lib.gm_forward_0_async(policy, _ptr(geom), P, int(degree), M, _ptr(bg), W, H, _ptr(means3D), _ptr(sh), _ptr(colors),
                                              _ptr(opacity), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                                              _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy),
                                              int(bool(prefiltered)), _ptr(radii), int(bool(debug)), stream.cuda_stream,
                                              count_host.data_ptr(), event.cuda_event))
    except Exception:
        if workspace is not None:
            workspace.release(h)
        raise
    h.args = dict(device=device, P=P, W=W, H=H, bg=bg, debug=int(bool(debug)), prefiltered=bool(prefiltered),
                  keep=(means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos))
    h.geom, h.img, h.color, h.radii, h.count_host, h.event = geom, img, color, radii, count_host, event
    return h


def rasterize_forward(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                      projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                      force_M=None, workspace=None, emission_policy=None):
    """RasterizeGaussiansCUDA of the reference bridge (rasterize_points.py:88-274): returns
    (num_rendered, color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer).  One host synchronisation (the
    instance count, reference rasterizer_impl.cu:411)."""
    return rasterize_forward_begin(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                                   tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug,
                                   workspace=workspace, emission_policy=emission_policy, force_M=force_M).finish()


def forward_deformed_begin(bg, tri, weights, packed, cov, pos, shs, opacity, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                           image_height, image_width, degree, campos, debug=False, workspace=None, want_deformed=False,
                           emission_policy=None, want_count=True, depth_plan=None):
    """Edit-loop frame, first half (gm_forward_0_deformed_async): mesh-driven deformation + rotated-direction SH colour +
    forward preprocess + depth order + instance count in one enqueue, no host synchronisation.  `packed` is
    deform.pack_mesh_state() of the frame.  Returns a PendingForward; .finish() completes the frame
    (gm_forward_1_geom) and returns (num_rendered, color, radii, geom, binning, img).  With want_deformed the handle
    also carries .deformed = (pos' [N,3], cov6 [N,6], rgb [N,3]).  want_count=False (loops that complete their frames with
    finish(sync_free=True)): the 4-byte copy of the instance count to the host is not enqueued either.
    cov: the rest covariances [N,3,3] / [N,9], or [N,6] from deform.pack_cov6() (bit-symmetric matrices: 12 bytes per Gaussian
    less to read, identical results).
    depth_plan (new_depth_plan(), one per view stream): see DepthPlan; complete such frames with finish(sync_free=True) and
    check() them, or the status of a refused frame goes unseen."""
    lib = _lib.lib()
    device = pos.device
    if device.type != "cuda":
        raise _lib.GmeshError("gaussianmesh_amd rasterizer needs tensors on a HIP (cuda) device; there is no CPU path")
    policy = _pol(emission_policy, image_width, image_height)
    P, M = pos.shape[0], shs.shape[1]
    if tri.dtype is not torch.int32 or not tri.is_contiguous():
        tri = tri.detach().contiguous().to(torch.int32)
    weights, packed, cov, pos, shs, opacity = (_prep(t, device) for t in (weights, packed, cov, pos, shs, opacity))   # None when empty
    bg, viewmatrix, projmatrix, campos = (_prep(t, device) for t in (bg, viewmatrix, projmatrix, campos))
    H, W = int(image_height), int(image_width)
    stream = _current_stream(device)
    _note_stream(stream)
    h = PendingForward(policy=policy, workspace=workspace, stream=stream)
    if workspace is not None:
        workspace.acquire(h)
    try:
        with _on(device):
            f = dict(dtype=torch.float32, device=device)
            color = torch.empty((3, H, W), **f)
            radii = torch.empty((P,), dtype=torch.int32, device=device)
            deformed = (torch.empty((P, 3), **f), torch.empty((P, 6), **f), torch.empty((P, 3), **f)) if want_deformed else None
            geom, img, count_host = _scratch(workspace, P, W, H, device)
            dp = [None, None, None] if deformed is None else [t.data_ptr() for t in deformed]
            event = _count_event(stream) if want_count else None
            if not want_count:
                count_host = None
            cov6 = cov is not None and cov.dim() == 2 and cov.shape[1] == 6       # deform.pack_cov6(): GM_STREAM_COV6
            direct = depth_plan is not None and depth_plan.primed and P > 0
            slab = None
            if direct:
                nbytes = lib.gm_depth_slab_bytes(P)
                slab = workspace.get("slab", nbytes, device) if workspace is not None else torch.empty((nbytes,), dtype=torch.uint8, device=device)

            def begin(direct_now, count_ptr, event_ptr):
                _lib.check(lib.gm_forward_0_deformed_stream_async(
                    policy, _ptr(geom), P, int(degree), M, W, H, _ptr(tri), _ptr(weights), _ptr(packed), _ptr(cov), _ptr(pos), _ptr(shs),
                    _ptr(opacity), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), dp[0], dp[1], dp[2],
                    _ptr(radii), int(bool(debug)), stream.cuda_stream, count_ptr, event_ptr, _ptr(slab) if direct_now else None,
                    None if depth_plan is None else depth_plan.buf.data_ptr(), (1 if direct_now else 0) | (2 if cov6 else 0)))

            begin(direct, None if count_host is None else count_host.data_ptr(), None if event is None else event.cuda_event)
            if depth_plan is not None and P > 0:
                depth_plan.primed = True               # (this frame, whichever path it took, leaves a table behind)
    except Exception:
        if workspace is not None:
            workspace.release(h)
        raise
    h.args = dict(device=device, P=P, W=W, H=H, bg=bg, debug=int(bool(debug)),
                  keep=(tri, weights, packed, cov, pos, shs, opacity, viewmatrix, projmatrix, campos))
    h.geom, h.img, h.color, h.radii, h.count_host, h.event, h.deformed = geom, img, color, radii, count_host, event, deformed
    if direct:
        h.direct = True

        def rebegin():
            depth_plan.refused += 1
            with _on(device):
                begin(False, None, None)
        h.rebegin = rebegin
    return h


def forward_deformed_batch(bg, tri, weights, packed_list, cov, pos, shs, opacity, cameras, image_height, image_width, degree, workspaces,
                           image_only=True, work_hint=None, emission_policy=None, debug=False):
    """K frames of one view stream in ONE launch chain (gm_forward_deformed_batch_async): the static cloud is read from HBM once for the
    batch and every stage is one launch over the K frames.  packed_list: the K gather tables (deform.mesh_rs_packed_batch);
    cameras: K dicts / objects with view, proj, campos (device tensors), tanx, tany; workspaces: K RasterWorkspace, one per frame, each with
    a learned capacity (frames of this stream completed through finish() before: the batch is sync-free only).  Returns K PendingForward
    handles in the state finish(sync_free=True) leaves them in: .result = (-1, color, radii, geom, binning, img); check() each of them -
    a frame whose instance count outgrew the batch's capacity is refused (image = background) and finish() renders it again, exactly,
    through the single-frame second half.  Each frame comes out bit for bit as from forward_deformed_begin(...).finish(sync_free=True)."""
    lib = _lib.lib()
    device = pos.device
    if device.type != "cuda":
        raise _lib.GmeshError("gaussianmesh_amd rasterizer needs tensors on a HIP (cuda) device; there is no CPU path")
    K = len(packed_list)
    if not (1 <= K <= _lib.GM_BATCH_MAX) or len(cameras) != K or len(workspaces) != K:
        raise ValueError("forward_deformed_batch: 1..%d frames, one camera and one workspace each" % _lib.GM_BATCH_MAX)
    if len({id(w_) for w_ in workspaces}) != K:
        raise ValueError("forward_deformed_batch: the frames of a batch need distinct workspaces")
    policy = _pol(emission_policy, image_width, image_height)
    P, M = pos.shape[0], shs.shape[1]
    if tri.dtype is not torch.int32 or not tri.is_contiguous():
        tri = tri.detach().contiguous().to(torch.int32)
    weights, cov, pos, shs, opacity, bg = (_prep(t, device) for t in (weights, cov, pos, shs, opacity, bg))
    H, W = int(image_height), int(image_width)
    stream = _current_stream(device)
    _note_stream(stream)
    cap = max(ws.capacity for ws in workspaces)
    if cap <= 0:
        raise _lib.GmeshError("forward_deformed_batch: the workspaces have no capacity yet - complete one frame of the stream through "
                              "forward_deformed_begin(...).finish() first (the batch is sync-free: it cannot size the binning buffers)")
    cov6 = cov is not None and cov.dim() == 2 and cov.shape[1] == 6
    get = lambda c, k: c[k] if isinstance(c, dict) else getattr(c, k)
    handles, frames, keep = [], (_lib.BatchFrame * K)(), []
    try:
        with _on(device), torch.cuda.stream(stream):
            nbin = lib.gm_binning_bytes(cap)
            for k in range(K):
                ws, c = workspaces[k], cameras[k]
                h = PendingForward(policy=policy, workspace=ws, stream=stream)
                ws.acquire(h)
                handles.append(h)
                ws.capacity = cap
                color = torch.empty((3, H, W), dtype=torch.float32, device=device)
                radii = torch.empty((P,), dtype=torch.int32, device=device)
                geom, img, _ = _scratch(ws, P, W, H, device)
                binning = ws.get("binning", nbin, device)
                view, proj, campos, packed = (_prep(t, device) for t in (get(c, "view"), get(c, "proj"), get(c, "campos"), packed_list[k]))
                keep.append((view, proj, campos, packed))
                f = frames[k]
                f.packed, f.viewmatrix, f.projmatrix, f.cam_pos = packed.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr()
                f.tan_fovx, f.tan_fovy = float(get(c, "tanx")), float(get(c, "tany"))
                f.geom_buffer, f.binning_buffer, f.image_buffer = geom.data_ptr(), binning.data_ptr(), img.data_ptr()
                f.out_color, f.radii, f.status_host = color.data_ptr(), radii.data_ptr(), ws.pinned_status().data_ptr()
                h.args = dict(device=device, P=P, W=W, H=H, bg=bg, debug=int(bool(debug)), keep=(tri, weights, cov, pos, shs, opacity, keep[-1]))
                h.geom, h.img, h.color, h.radii, h.count_host, h.event, h.deformed, h.binning = geom, img, color, radii, None, None, None, binning
                h.image_only, h.work_hint = bool(image_only), work_hint
            _lib.check(lib.gm_forward_deformed_batch_async(policy, K, frames, P, int(degree), M, W, H, _ptr(tri), _ptr(weights), _ptr(cov), _ptr(pos), _ptr(shs),
                                                           _ptr(opacity), _ptr(bg), cap, (1 if image_only else 0) | (2 if cov6 else 0),
                                                           None if work_hint is None else work_hint.data_ptr(), int(bool(debug)), stream.cuda_stream))
            for h in handles:
                h.status_event = h.workspace.status_event()
                h.status_event.record(stream)
                h.result = (-1, h.color, h.radii, h.geom, h.binning, h.img)
    except Exception:
        for h in handles:
            h.workspace.release(h)
        raise
    return handles


def rasterize_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                       tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geom, num_rendered, binning, img, debug,
                       emission_policy=None, skip_intermediates=False, want_conic=False, sh_step=None):
    """RasterizeGaussiansBackwardCUDA of the reference bridge (rasterize_points.py:276-401): returns
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).
    emission_policy: the policy the forward that filled geom / binning / img ran under.
    skip_intermediates (the autograd operator): dL_dcolors when the colours come from SH rows and dL_dcov3D when the covariances come
    from scale / rotation are not computed into memory (returned as None), nor is the internal dL/dconic.
    sh_step (an ShStep, see below): the Adam step of the SH rows happens inside the backward (gm_backward_sh_step); dL_dsh, dL_dcolors and
    dL_dcov3D (with scales) come back as None.
    want_conic (tests): a ninth return value, dL_dconic [P,2,2] (slots [0,0], [0,1], [1,1] used: the blend stage's output that the
    reference keeps internal, rasterize_points.py:305)."""
    lib = _lib.lib()
    device = means3D.device
    P = means3D.shape[0]
    f = dict(dtype=torch.float32, device=device)
    M_in = sh.shape[1] if (sh is not None and sh.numel() > 0 and sh.dim() == 3) else 0
    if P == 0:                                  # empty cloud: zero-size gradients, nothing to launch
        z = lambda *shape: torch.zeros(shape, **f)
        return z(0, 3), z(0, 3), z(0, 1), z(0, 3), z(0, 6), z(0, M_in, 3), z(0, 3), z(0, 4)
    means3D = _prep(means3D, device)
    sh, colors, scales, rotations, cov3D_precomp = (_prep(t, device) for t in (sh, colors, scales, rotations, cov3D_precomp))
    bg, viewmatrix, projmatrix, campos = (_prep(t, device) for t in (bg, viewmatrix, projmatrix, campos))
    dpix = _prep(dL_dout_color, device)
    H, W = dpix.shape[1], dpix.shape[2]
    M = sh.shape[1] if sh is not None else 0
    if sh_step is not None:
        if sh is None or M != 16 or colors is not None or sh.data_ptr() != sh_step.param.data_ptr():
            raise _lib.GmeshError("rasterize_backward(sh_step=...): the step's parameter must be the [P,16,3] shs operand of this pass")
        with _on(device):
            dmeans2D = torch.empty((P, 3), **f); dopac = torch.empty((P, 1), **f); dmeans3D = torch.empty((P, 3), **f)
            dcov3D = None if scales is not None else torch.empty((P, 6), **f)
            dscales = torch.empty((P, 3), **f) if scales is not None else None
            drots = torch.empty((P, 4), **f) if scales is not None else None
            _lib.check(lib.gm_backward_sh_step(_pol(emission_policy, W, H), P, int(degree), M, int(num_rendered), _ptr(bg), W, H, _ptr(means3D), _ptr(sh),
                                               _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix),
                                               _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geom), _ptr(binning), _ptr(img), _ptr(dpix),
                                               _ptr(dmeans2D), _ptr(dopac), _ptr(dmeans3D), _ptr(dcov3D), _ptr(dscales), _ptr(drots), int(sh_step.rows),
                                               sh_step.exp_avg.data_ptr(), sh_step.exp_avg_sq.data_ptr(), float(sh_step.lr_dc), float(sh_step.lr_rest),
                                               float(sh_step.betas[0]), float(sh_step.betas[1]), float(sh_step.eps), int(sh_step.step), int(bool(debug)),
                                               _stream(device)))
        sh_step.applied = True
        return dmeans2D, None, dopac, dmeans3D, dcov3D, None, dscales, drots
    with _on(device):
        skip = bool(skip_intermediates)
        dmeans2D = torch.empty((P, 3), **f); dopac = torch.empty((P, 1), **f); dmeans3D = torch.empty((P, 3), **f)
        dconic = None if skip else torch.empty((P, 2, 2), **f)
        dcolors = None if (skip and sh is not None) else torch.empty((P, 3), **f)
        dcov3D = None if (skip and scales is not None) else torch.empty((P, 6), **f)
        dsh = torch.empty((P, M, 3), **f) if sh is not None else None
        dscales = torch.empty((P, 3), **f) if scales is not None else None
        drots = torch.empty((P, 4), **f) if scales is not None else None
        _lib.check(lib.gm_backward_p(_pol(emission_policy, W, H), P, int(degree), M, int(num_rendered), _ptr(bg), W, H, _ptr(means3D), _ptr(sh),
                                     _ptr(colors), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                                     _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geom),
                                     _ptr(binning), _ptr(img), _ptr(dpix), _ptr(dmeans2D), _ptr(dconic), _ptr(dopac),
                                     _ptr(dcolors), _ptr(dmeans3D), _ptr(dcov3D), _ptr(dsh), _ptr(dscales), _ptr(drots),
                                     int(bool(debug)), _stream(device)))
    if want_conic:
        return dmeans2D, dcolors, dopac, dmeans3D, dcov3D, dsh, dscales, drots, dconic
    return dmeans2D, dcolors, dopac, dmeans3D, dcov3D, dsh, dscales, drots


class ShStep:
    """The Adam step of the SH rows, to be applied INSIDE the next backward pass of the rasterizer whose `shs` operand is `param`'s storage
    (gm_backward_sh_step): `param` [rows,16,3] (a leaf; the operand may continue with frozen rows behind it), exp_avg / exp_avg_sq
    [rows,16,3], lr_dc for coefficient 0 and lr_rest for the others (the reference's "f_dc" / "f_rest" groups), `step` = the step being
    taken (bias correction).  Make it current for the calling thread around the FORWARD and the backward (the operator's forward picks it up:
    backward() runs on the autograd engine's thread):

        with ShStep(...) as ss:
            loss = f(render(...)); loss.backward()
        ss.applied      # True: the operator took the fused route (the leaf's .grad stays None: nothing left for the optimizer to do)

    One step per ShStep object: a second backward inside the block takes the ordinary route.  The backward of a refused sync-free forward
    leaves parameter and moments untouched on the device (applied is True all the same: the caller repeats the iteration with a new one)."""

    def __init__(self, param, exp_avg, exp_avg_sq, lr_dc, lr_rest, betas, eps, step):
        if param.dim() != 3 or param.shape[1] != 16 or param.shape[2] != 3 or tuple(exp_avg.shape) != tuple(param.shape) or tuple(exp_avg_sq.shape) != tuple(param.shape):
            raise ValueError("ShStep: parameter and moments must be [rows,16,3]")
        for t in (param, exp_avg, exp_avg_sq):
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError("ShStep: contiguous float32 tensors expected")
        self.param, self.exp_avg, self.exp_avg_sq, self.rows = param, exp_avg, exp_avg_sq, int(param.shape[0])
        self.lr_dc, self.lr_rest, self.betas, self.eps, self.step = float(lr_dc), float(lr_rest), betas, float(eps), int(step)
        self.applied = False

    def __enter__(self):
        stack = getattr(_sync_free_tls, "sh_steps", None)
        if stack is None:
            stack = _sync_free_tls.sh_steps = []
        stack.append(self)
        return self

    def __exit__(self, *exc):
        _sync_free_tls.sh_steps.pop()
        return False


def current_sh_step():
    stack = getattr(_sync_free_tls, "sh_steps", None)
    return stack[-1] if stack else None


def mark_visible(means3D, viewmatrix, projmatrix):
    """rasterize_points.py:39-58 mark_visible -> bool [P]."""
    lib = _lib.lib()
    device = means3D.device
    means3D, viewmatrix, projmatrix = (_prep(t, device) for t in (means3D, viewmatrix, projmatrix))
    P = 0 if means3D is None else means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.uint8, device=device)
    if P:
        with _on(device):
            _lib.check(lib.gm_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), _ptr(present), _stream(device)))
    return present.bool()


_WORKSPACES = {}


def _snapshot(path, args):
    """torch.save of a failing call's arguments (tensors moved to the host when the device still answers)"""
    def host(v):
        try:
            return v.detach().cpu() if isinstance(v, torch.Tensor) else v
        except Exception:
            return None
    try:
        torch.save({k: host(v) for k, v in args.items()}, path)
    except Exception:
        pass


def _shared_workspace(device):
    """inference scratch of the autograd operator, one per (device, stream, thread)"""
    import threading
    key = (device, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = RasterWorkspace()
    return ws


# Sync-free training forward: the autograd operator never waits for the instance count.  The binning buffer of an iteration
# is sized from the largest count seen so far (x growth); the forward's status words are checked by SyncFreeState.verify() -
# typically after backward() has been enqueued, so the host never idles the GPU - and an iteration whose count outgrew its
# buffer (image = background, gradients = 0) is reported so the caller can redo it.
#
# The state (capacity per device, the forwards not yet verified) belongs to its OWNER - a train.Trainer holds one - and the
# operator sees it only while the owner has made it current for the calling THREAD (`with state:`): two trainers in one
# process, or two threads, do not share a capacity guess or an unverified list.  Nothing here is module-global.
_SYNC_FREE_MAX_UNCHECKED = 64      # unverified sync-free forwards (each pins its geometry / binning / image buffers)


class SyncFreeState:
    """Per-owner state of the sync-free training forward.

        state = SyncFreeState()
        with state:                      # forwards that need a gradient, issued by this thread inside the block, are sync-free
            loss = f(render(...)); loss.backward()
            ok = state.verify()          # False: the instance count outgrew the buffer; redo the iteration (capacity was raised)

    capacity[device]: instances the next binning buffer holds (0 / absent: the next forward takes the exact-count path and
    seeds the guess).  `enabled = False` keeps the block on the exact path (and keeps learning the capacity)."""

    def __init__(self, growth=1.3, enabled=True):
        self.growth = float(growth)
        self.enabled = bool(enabled)
        self.capacity = {}
        self.unchecked = []

    def __enter__(self):
        stack = getattr(_sync_free_tls, "stack", None)
        if stack is None:
            stack = _sync_free_tls.stack = []
        stack.append(self)
        return self

    def __exit__(self, *exc):
        stack = _sync_free_tls.stack
        if not stack or stack[-1] is not self:
            raise _lib.GmeshError("SyncFreeState: blocks must be exited in the order they were entered")
        stack.pop()
        self.unchecked.clear()           # handles left unverified (an exception inside the block) must not pin their buffers
        return False

    def note_count(self, device, num_rendered):
        self.capacity[device] = max(self.capacity.get(device, 0), int(num_rendered * self.growth) + 4096)

    def scale_capacity(self, factor, extra=4096):
        """the cloud grew by `factor` rows (a topology change): scale the guesses with it"""
        for k in list(self.capacity):
            self.capacity[k] = int(self.capacity[k] * factor) + extra

    def verify(self):
        """Wait for the status of every sync-free forward issued under this state since the last call (the status words were
        written by the forwards' own blend kernels, long before this is normally called).  Returns True when all of them fitted
        their binning buffers; on False the capacity guess has been raised and the caller should repeat the iteration (its image
        was the background)."""
        ok = True
        for h in self.unchecked:
            fitted, nr = h.check()
            self.note_count(h.args["device"], nr)
            ok = ok and fitted
        self.unchecked.clear()
        return ok


import threading as _threading
_sync_free_tls = _threading.local()


def current_sync_free():
    """The SyncFreeState the calling thread is inside of, or None."""
    stack = getattr(_sync_free_tls, "stack", None)
    return stack[-1] if stack else None


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                force_M):
        rs = raster_settings
        if means3D.device.type != "cuda":
            raise _lib.GmeshError("gaussianmesh_amd rasterizer needs tensors on a HIP (cuda) device; there is no CPU path")
        needs_grad = any(ctx.needs_input_grad)
        ws = None if needs_grad else _shared_workspace(means3D.device)   # inference: reuse scratch
        policy = get_default_emission_policy(rs.image_width, rs.image_height)
        sf = current_sync_free() if needs_grad else None
        cap = sf.capacity.get(means3D.device, 0) if (sf is not None and sf.enabled) else 0
        try:
            h = rasterize_forward_begin(rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                                        rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                                        rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, workspace=ws, emission_policy=policy,
                                        force_M=force_M)
            if cap > 0:
                num_rendered, color, radii, geom, binning, img = h.finish(sync_free=True, capacity=cap, work_hint=rs.work_hint, exact_exponent=True)
                num_rendered = cap                        # the binning layout is that of the capacity
                if len(sf.unchecked) >= _SYNC_FREE_MAX_UNCHECKED:
                    raise _lib.GmeshError("sync-free training: %d forwards were issued without SyncFreeState.verify(); every "
                                          "unverified forward keeps its scratch buffers alive - call verify() once per iteration"
                                          % _SYNC_FREE_MAX_UNCHECKED)
                sf.unchecked.append(h)
            else:
                # image_only: no backward will follow; exact_exponent: one will, and takes the decisions this forward took
                num_rendered, color, radii, geom, binning, img = h.finish(image_only=not needs_grad, work_hint=rs.work_hint, exact_exponent=needs_grad)
                if sf is not None:
                    sf.note_count(means3D.device, num_rendered)
        except Exception:
            if rs.debug:       # the reference's debugging aid (diff_gaussian_rasterizater/__init__.py:61-67): keep the failing call's inputs
                _snapshot("snapshot_fw.dump", dict(bg=rs.bg, means3D=means3D, colors_precomp=colors_precomp, opacities=opacities, scales=scales,
                                                   rotations=rotations, scale_modifier=rs.scale_modifier, cov3Ds_precomp=cov3Ds_precomp,
                                                   viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                                                   image_height=rs.image_height, image_width=rs.image_width, sh=sh, sh_degree=rs.sh_degree,
                                                   campos=rs.campos, prefiltered=rs.prefiltered))
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise
        # an ShStep current for the calling thread whose parameter is this pass's shs operand rides on the context: backward() runs on the
        # autograd engine's own thread, where the caller's thread-local state is not visible
        ss = current_sh_step() if needs_grad else None
        if ss is not None and (sh is None or sh.numel() == 0 or sh.dim() != 3 or sh.shape[1] != 16 or sh.data_ptr() != ss.param.data_ptr()
                               or (colors_precomp is not None and colors_precomp.numel() > 0)):
            ss = None
        ctx.sh_step = ss
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.emission_policy = policy
        ctx.opacity_shape = opacities.shape
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        ss = ctx.sh_step
        if ss is not None and ss.applied:
            ss = None                            # one step per ShStep: a second backward takes the ordinary route
        try:
            g2d, gcol, gop, g3d, gcov, gsh, gsc, grot = rasterize_backward(
                rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, geom, ctx.num_rendered,
                binning, img, rs.debug, ctx.emission_policy, skip_intermediates=True, sh_step=ss)
        except Exception:
            if rs.debug:       # diff_gaussian_rasterizater/__init__.py:102-108
                _snapshot("snapshot_bw.dump", dict(bg=rs.bg, means3D=means3D, radii=radii, colors_precomp=colors_precomp, scales=scales,
                                                   rotations=rotations, scale_modifier=rs.scale_modifier, cov3Ds_precomp=cov3Ds_precomp,
                                                   viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                                                   grad_out_color=grad_out_color, sh=sh, sh_degree=rs.sh_degree, campos=rs.campos,
                                                   num_rendered=ctx.num_rendered))
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
            raise
        has = lambda t: t is not None and t.numel() > 0
        return (g3d, g2d, gsh if (has(sh) and gsh is not None) else None, gcol if has(colors_precomp) else None, gop.reshape(ctx.opacity_shape),
                gsc if has(scales) else None, grot if has(rotations) else None, gcov if has(cov3Ds_precomp) else None,
                None, None)


class GaussianRasterizer(nn.Module):
    _force_M = None

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = lambda: torch.empty(0, device=means3D.device)
        shs = e() if shs is None else shs
        colors_precomp = e() if colors_precomp is None else colors_precomp
        scales = e() if scales is None else scales
        rotations = e() if rotations is None else rotations
        cov3D_precomp = e() if cov3D_precomp is None else cov3D_precomp
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                         raster_settings, self._force_M)

    execute = forward      # Jittor spelling used by the reference (nn.Module.execute)


class NewGaussianRasterizer(GaussianRasterizer):
    """Edit-tool variant: identical op, SH stride fixed at 16 coefficients (rasterize_points_deformed.py:157,230)."""
    _force_M = 16
