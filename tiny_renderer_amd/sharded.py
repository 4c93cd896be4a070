"""Screen-band sharding of one frame over the GPUs of a node (north_star).

One process per GPU (torch.distributed, backend "nccl" = RCCL).  Rank r renders output rows
tr_band_rows(height, world, r) -- the reference's own clamp rectangle (scene.rs:236-239) cut to
the band, so a band is bit-identical to the same rows of the single-GPU frame -- into its slice
of a full-size frame tensor, and the bands are exchanged with one in-place all-gather per frame.
Frames are double-buffered: the exchange of frame f runs on a second stream, ordered by events,
under the render of frame f + 1.  Depth passes of shadow / occlusion are rendered for the whole
frame on every rank (their lookups are in light space, shader.rs:774-778): no second collective.

`ShardedScene` mirrors the reference's `Scene` methods (clear / set_light_direction / set_camera /
render / get_frame_buffer) so that the headless CLI can drive either; bench.py spells the same
loop out because it times its parts.
"""
import os
import socket
import subprocess
import sys

from .scene import Scene, band_rows


def rank_environment():
    """Environment of a rank process: the package's parent directory on PYTHONPATH (the ranks are started as
    `-m tiny_renderer_amd.cli`, whatever the caller's working directory is), dmabuf IPC for RCCL."""
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    return env


def rank_command(n_gpus, module, argv, port):
    """The command that starts `n_gpus` ranks of `python -m module argv...`, one per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", module] + list(argv)


def launch_ranks(n_gpus, module, argv):
    """Starts `n_gpus` rank processes of `python -m module` (one per GPU) with torch.distributed.run and returns
    their exit code.  Must be called from a process that has not touched a GPU (no torch.cuda call):
    the ranks are children, nothing is exec'ed over an initialised process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = rank_command(n_gpus, module, argv, port)
    sys.stderr.write("starting %d ranks: %s\n" % (n_gpus, " ".join(cmd)))
    return subprocess.run(cmd, env=rank_environment()).returncode


def worst_status(code, device=None):
    """The smallest (most severe: error codes are negative) status over all ranks, one tiny all-reduce.  Decisions
    that change how many collectives a rank issues -- rendering a frame again after its bins overflowed, giving
    up on an error -- must be taken together, or the ranks fall out of step and the next collective never
    completes."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(code)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([int(code)], dtype=torch.int32,
                     device=("cuda:%d" % (torch.cuda.current_device() if device is None else device)) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def any_rank(flag, device=None):
    """True on every rank if `flag` is true on any."""
    return worst_status(-1 if flag else 0, device) != 0


class _DeviceBytes:
    """A device buffer of the library's as something torch can wrap (CUDA array interface): the frame slots of a
    PeerExchange become tensors, so that every frame -- whoever owns the memory -- is read the same way."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class _TorchGather:
    """exchange="torch": frame tensors of torch's, one in-place all_gather_into_tensor per frame (backend "nccl" = RCCL)."""
    name = "torch"

    def __init__(self, torch, dist, n_slots, frame_bytes, rank, world, device):
        if dist.get_backend() != "nccl":
            raise ValueError('exchange="torch" moves the bands with torch.distributed collectives on device tensors: the '
                             'process group must use the "nccl" (RCCL) backend')
        self._dist, self.rank = dist, rank
        self.tensors = [torch.zeros(frame_bytes, dtype=torch.uint8, device="cuda:%d" % device) for _ in range(n_slots)]

    def gather(self, slot, offset, nbytes, scene, stream):
        t = self.tensors[slot]
        self._dist.all_gather_into_tensor(t, t[offset:offset + nbytes])   # (on torch's current stream: the caller's `with`)

    def bytes_sent(self):
        return None

    def status(self):
        return 0

    def close(self):
        self.tensors = []


class _LibraryGather:
    """exchange="peer" / "peer-sparse" / "rccl": the library's own exchange (tr_exchange_*, PeerExchange below) owns the
    frame slots; torch.distributed (any backend, gloo is enough) only carries the connection records."""

    def __init__(self, torch, dist, n_slots, frame_bytes, rank, world, device, kind, ranges):
        self.name = kind
        self.sparse = kind == "peer-sparse"
        self.x = PeerExchange(n_slots, frame_bytes, rank, world, device, backend="rccl" if kind == "rccl" else "peer")
        self.x.set_ranges(*ranges)   # (dense peer exchange: pull -- nothing but flags is written into a peer's memory)
        self.tensors = [torch.as_tensor(_DeviceBytes(self.x.frame_ptr(b), frame_bytes), device="cuda:%d" % device)
                        for b in range(n_slots)]

    def gather(self, slot, offset, nbytes, scene, stream):
        if self.sparse:
            self.x.all_gather_tiles(slot, scene.band_tiles(self.x.frame_ptr(slot)), stream.cuda_stream)
        else:
            self.x.all_gather(slot, offset, nbytes, stream.cuda_stream)

    def bytes_sent(self):
        return self.x.bytes_sent()

    def status(self):
        return self.x.status()

    def close(self):
        self.tensors = []
        self.x.close()


EXCHANGES = ("torch", "rccl", "peer", "peer-sparse")


class ShardedScene:
    """Scene::new(...) for one rank of an initialised process group; every rank ends up with the whole frame.

    exchange: how the bands travel --
      "torch"        torch.distributed's all_gather_into_tensor (RCCL; the default, north_star's collective),
      "rccl"         the library's own RCCL communicator (tr_exchange_create_backend(TR_EXCHANGE_RCCL): no torch in the data path),
      "peer"         the library's peer transport: bands pulled out of the peers' IPC-mapped slots by the DMA engines,
      "peer-sparse"  ... tile by tile, tiles that are the cleared colour on both sides stay home (k_push_tiles).
    The render side -- band scene, double-buffered frames, groups, overflow repair, statuses reduced over the ranks -- is
    the same for all four.  RCCL wants one device per rank; the peer transports also run with several ranks on one GPU
    (how the multi-rank path is rehearsed on a one-GPU box: tests/sharded_ranks_worker.py)."""

    def __init__(self, width, height, mesh, textures, shader_pipeline_name, *, device=None, exchange="torch", **scene_kw):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.width, self.height = int(width), int(height)
        self.band = band_rows(self.height, self.world, self.rank)
        bands = [band_rows(self.height, self.world, r) for r in range(self.world)]
        if len({b[1] - b[0] for b in bands}) != 1:
            raise ValueError("frame height %d must divide by the number of GPUs %d (in-place all-gather)"
                             % (self.height, self.world))
        if exchange not in EXCHANGES:
            raise ValueError("exchange must be one of %s" % (EXCHANGES,))
        if device is None:
            device = torch.cuda.current_device()
        self._device = device
        self._render = torch.cuda.Stream(device=device)
        self._comm = torch.cuda.Stream(device=device)
        assert self._render.cuda_stream != 0  # tr_options.stream = NULL would mean a library-owned stream
        n_all = self.width * self.height * 3
        row = self.width * 3
        self._band_bytes = (self.band[1] - self.band[0]) * row
        self._band_offset = self.band[0] * row
        # frame slots: 0, 1 = the per-frame protocol's double buffer; then two sets of `frames per launch` for render_frames
        # (made when the first group is rendered: _group_slots)
        self._exchange_kind = exchange
        self._ranges = ([b[0] * row for b in bands], [(b[1] - b[0]) * row for b in bands])
        self._gather = self._make_gather(2)
        self._fbs = self._gather.tensors[:2]
        self._rendered = [torch.cuda.Event() for _ in range(2)]
        self._gathered = [torch.cuda.Event() for _ in range(2)]
        torch.cuda.synchronize(device)
        # (the frame tensors are written by this scene and by the exchange of OTHER ranks' rows only: the scene may
        # trust what it remembers about its own rows of them)
        self._scene = Scene(width, height, mesh, textures, shader_pipeline_name, device=device,
                            stream=self._render.cuda_stream, frame_buffer_device=self._fbs[0].data_ptr(),
                            band_rows=self.band, trust_frame_buffers=True, **scene_kw)
        self._slot = 1          # the first frame goes to slot 0
        self._cleared = True    # Scene::new leaves cleared (zero-filled) targets
        self._used = [False, False]
        self._last_was_cleared = True
        self._last_tensor = None   # where the newest frame is when it came from render_frames (else _fbs[_slot])
        self._last_group = None    # (frames, set) of the newest group: rendered again if its bins overflowed
        self._timing = None        # enable_timing: [(render start, render end, gather start, gather end, frames)]

    def enable_timing(self, on=True):
        """Device times of this rank's renders and exchanges from now on (timing events on the two streams: a few
        microseconds per frame or group, so only for a measuring leg); timings() reads them."""
        self._timing = [] if on else None

    def _stamp(self, stream):
        e = self._torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def timings(self):
        """{"render_us", "gather_us"}: medians per FRAME over what was issued since enable_timing (waits for it)."""
        import numpy as np
        if not self._timing:
            return {"render_us": None, "gather_us": None}
        self._torch.cuda.synchronize()
        r = [a.elapsed_time(b) * 1e3 / n for a, b, _, _, n in self._timing]
        g = [c.elapsed_time(d) * 1e3 / n for _, _, c, d, n in self._timing]
        return {"render_us": round(float(np.median(r)), 2), "gather_us": round(float(np.median(g)), 2)}

    def _make_gather(self, n_slots):
        torch, dist = self._torch, self._dist
        n_all = self.width * self.height * 3
        if self._exchange_kind == "torch":
            return _TorchGather(torch, dist, n_slots, n_all, self.rank, self.world, self._device)
        return _LibraryGather(torch, dist, n_slots, n_all, self.rank, self.world, self._device, self._exchange_kind, self._ranges)

    # --- the reference's methods -------------------------------------------------------------
    def clear(self):
        self._cleared = True
        self._scene.clear()

    def set_light_direction(self, v):
        self._scene.set_light_direction(v)

    def set_camera(self, look_from, look_at, up):
        self._scene.set_camera(look_from, look_at, up)

    def render(self):
        """Renders this rank's band and starts its exchange.  The reference's per-frame protocol is
        clear -> set_* -> render: a cleared frame moves on to the other frame tensor (so that the
        previous frame's exchange can still be running); a render WITHOUT a clear accumulates into the
        same tensor, as `render` does upstream (scene.rs:151), and is exchanged again."""
        torch, dist = self._torch, self._dist
        if self._cleared:
            self._slot ^= 1
        b = self._slot
        with torch.cuda.stream(self._render):
            if self._used[b]:
                self._render.wait_event(self._gathered[b])  # the slot's previous exchange has finished
            t0 = self._stamp(self._render) if self._timing is not None else None
            self._scene.set_frame_buffer_device(self._fbs[b].data_ptr())
            self._scene.render()  # a caller's stream holds the frame's kernels when render() returns
            t1 = self._stamp(self._render) if self._timing is not None else None
            self._rendered[b].record(self._render)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(self._rendered[b])
            t2 = self._stamp(self._comm) if self._timing is not None else None
            self._gather.gather(b, self._band_offset, self._band_bytes, self._scene, self._comm)
            if self._timing is not None:
                self._timing.append((t0, t1, t2, self._stamp(self._comm), 1))
            self._gathered[b].record(self._comm)
        self._used[b] = True
        self._last_was_cleared = self._cleared
        self._cleared = False
        self._last_tensor = None
        self._last_group = None

    def render_frames(self, frames):
        """Many frames per call (Scene.render_frames on every rank); afterwards only the LAST frame is exposed
        (get_frame_buffer): each rank renders its band of a group of
        frames by one launch of each kernel into a set of frame tensors of the group's size, and the bands
        of the group are exchanged frame by frame on the second stream while the next group renders into
        the other set.  Frame i is what clear(); set_light_direction; set_camera; render() gives; afterwards
        the last frame is the one get_frame_buffer() returns.  Collective: all ranks pass the same frames."""
        import numpy as np
        torch, dist = self._torch, self._dist
        frames = np.ascontiguousarray(frames, np.float32).reshape(-1, 12)
        if len(frames) == 0:
            return
        G = self._scene.frames_per_launch
        if not hasattr(self, "_gsets"):
            # two sets of G frame slots: an exchange of its own for them (the library's exchanges own their slots: up to 64)
            if 2 * G > 64 and self._exchange_kind != "torch":
                raise ValueError("frames_per_launch %d: the library's exchange holds 64 frame slots (two sets of 32)" % G)
            self._ggather = self._make_gather(2 * G)
            self._gsets = [self._ggather.tensors[:G], self._ggather.tensors[G:2 * G]]
            self._grendered = [torch.cuda.Event() for _ in range(2)]
            self._ggathered = [torch.cuda.Event() for _ in range(2)]
            self._gused = [False, False]
            self._gset = 1
        for i0 in range(0, len(frames), G):
            g = min(G, len(frames) - i0)
            self._gset ^= 1
            b = self._gset
            with torch.cuda.stream(self._render):
                if self._gused[b]:
                    self._render.wait_event(self._ggathered[b])   # the set's previous exchange has finished
                t0 = self._stamp(self._render) if self._timing is not None else None
                self._scene.render_frames(frames[i0:i0 + g], [t.data_ptr() for t in self._gsets[b][:g]])
                t1 = self._stamp(self._render) if self._timing is not None else None
                self._grendered[b].record(self._render)
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(self._grendered[b])
                t2 = self._stamp(self._comm) if self._timing is not None else None
                for j in range(g):
                    self._ggather.gather(b * G + j, self._band_offset, self._band_bytes, self._scene, self._comm)
                if self._timing is not None:
                    self._timing.append((t0, t1, t2, self._stamp(self._comm), g))
                self._ggathered[b].record(self._comm)
            self._gused[b] = True
            self._last_group = (frames[i0:i0 + g].copy(), b)
        self._last_tensor = self._gsets[self._gset][g - 1]
        q = frames[-1]
        self._scene.set_light_direction(q[0:3])
        self._scene.set_camera(q[3:6], q[6:9], q[9:12])
        self._cleared = False
        self._last_was_cleared = True

    def sync(self):
        """Waits for the frames issued so far on every rank (collective: all ranks call it together).
        A band whose triangle bins overflowed was exchanged truncated -- the scene renders on a caller's
        stream, where the library reports that instead of repairing it behind the consumer's back
        (TR_E_BIN_OVERFLOW) -- so the last frame is rendered and exchanged again, by ALL ranks, now that
        the bins have grown."""
        from ._lib import TinyRendererError, TR_E_BIN_OVERFLOW
        for attempt in range(4):
            self._torch.cuda.synchronize()
            message = ""
            try:
                status = self._scene.sync()
                for gth in (self._gather, getattr(self, "_ggather", None)):
                    if gth is not None:
                        gth.status()   # (a peer's band that did not arrive: TR_E_EXCHANGE)
            except TinyRendererError as e:
                status, message = e.code, str(e)
            # every rank learns the most severe status of any rank and acts on THAT: a rank that raised on its own
            # (a lookup out of range in one band only) would leave the others waiting in the next collective
            worst = worst_status(status)
            if worst != TR_E_BIN_OVERFLOW:
                if worst < 0:
                    raise TinyRendererError(worst, message if status == worst else "raised on another rank")
                return status
            if not self._last_was_cleared:
                raise TinyRendererError(TR_E_BIN_OVERFLOW, "bins overflowed during an accumulating render: clear and render again")
            if self._last_group is not None:
                again, b = self._last_group
                self._gset = b ^ 1       # ... into the same set of frame tensors
                self.render_frames(again)
                continue
            self._scene.clear()          # same light and camera: the scene still holds them
            self._cleared = True
            self._slot ^= 1              # ... and the same frame tensor
            self.render()
        raise TinyRendererError(TR_E_BIN_OVERFLOW, "triangle bins kept overflowing")

    def get_frame_buffer(self):
        """The whole frame (all bands), [H, W, 3] uint8, row 0 = top.  Collective, like sync()."""
        status = self.sync()
        if status != 0:
            raise RuntimeError("device status %d" % status)
        t = self._last_tensor if self._last_tensor is not None else self._fbs[self._slot]
        return t.cpu().numpy().reshape(self.height, self.width, 3)

    def last_frame_tensor(self):
        """The frame tensor the newest frame is (being) exchanged into -- not collective: valid once a sync() of all
        ranks has returned."""
        return self._last_tensor if self._last_tensor is not None else self._fbs[self._slot]

    @property
    def scene(self):
        """This rank's band scene (profiling, flush)."""
        return self._scene

    def exchange_bytes_sent(self):
        """Bytes this rank's exchanges have pushed / offered to its peers so far (None: torch's collective does not say)."""
        n = [g.bytes_sent() for g in (self._gather, getattr(self, "_ggather", None)) if g is not None]
        return None if any(v is None for v in n) else sum(n)

    def close(self):
        self._torch.cuda.synchronize()
        self._scene.close()
        if self.world > 1:
            self._dist.barrier()   # nobody unmaps a slot a peer may still be reading
        for g in (self._gather, getattr(self, "_ggather", None)):
            if g is not None:
                g.close()




class PeerExchange:
    """The library's own frame exchange (tr_exchange_*, include/tiny_renderer.h): every rank's frame
    slots mapped into every other rank through HIP IPC, bands pushed with concurrent DMA-engine copies.
    `dist` (any initialised torch.distributed backend, gloo is enough) only carries the 256-byte
    connection records."""

    def __init__(self, n_slots, frame_bytes, rank, world, device, backend="peer"):
        """backend: "peer" (IPC-mapped slots, DMA-engine band copies) or "rccl" (the library's own RCCL communicator:
        one in-place ncclAllGather per call; `dist` still only carries the records)."""
        import ctypes as C
        import torch.distributed as dist
        from ._lib import check, load_library, TR_EXCHANGE_HANDLE_BYTES, TR_EXCHANGE_PEER, TR_EXCHANGE_RCCL
        L = load_library()
        self._L, self._check = L, check
        self.n_slots, self.frame_bytes, self.rank, self.world = n_slots, frame_bytes, rank, world
        h = C.c_void_p()
        check(L.tr_exchange_create_backend(device, world, rank, n_slots, frame_bytes,
                                           {"peer": TR_EXCHANGE_PEER, "rccl": TR_EXCHANGE_RCCL}[backend], C.byref(h)))
        self._h = h
        mine = C.create_string_buffer(TR_EXCHANGE_HANDLE_BYTES)
        check(L.tr_exchange_export(h, mine))
        records = [None] * world
        if world > 1:
            dist.all_gather_object(records, bytes(mine.raw))
        else:
            records[0] = bytes(mine.raw)
        check(L.tr_exchange_connect(h, b"".join(records)))
        if world > 1:
            dist.barrier()   # nobody pushes before everybody has mapped everybody

    def frame_ptr(self, slot):
        return self._L.tr_exchange_frame(self._h, slot)

    def all_gather(self, slot, offset, nbytes, hip_stream):
        self._check(self._L.tr_exchange_all_gather(self._h, slot, offset, nbytes, hip_stream))

    def set_ranges(self, offsets, nbytes):
        """tr_exchange_set_ranges: every rank's byte range of a frame (the dense peer exchange then pulls)."""
        import ctypes as C
        n = self.world
        off = (C.c_size_t * n)(*[int(v) for v in offsets])
        siz = (C.c_size_t * n)(*[int(v) for v in nbytes])
        self._check(self._L.tr_exchange_set_ranges(self._h, off, siz))

    def all_gather_tiles(self, slot, tiles, hip_stream):
        """The sparse exchange (tr_exchange_all_gather_tiles): `tiles` = Scene.band_tiles(self.frame_ptr(slot)) of the
        scene that rendered this rank's band into the slot; tiles that are the cleared colour on both sides stay home."""
        import ctypes as C
        self._check(self._L.tr_exchange_all_gather_tiles(self._h, slot, C.byref(tiles), hip_stream))

    def read(self, slot, height, width):
        import numpy as np
        out = np.empty((height, width, 3), np.uint8)
        self._check(self._L.tr_exchange_read(self._h, slot, out.ctypes.data, out.nbytes))
        return out

    def status(self):
        return self._check(self._L.tr_exchange_status(self._h))

    def bytes_sent(self):
        return int(self._L.tr_exchange_bytes_sent(self._h))

    def close(self):
        """Collective when world > 1 (every rank closes its end): unmap the peers' slots, wait until everybody has
        (nobody frees memory a peer still has mapped), then free."""
        if self._h:
            self._check(self._L.tr_exchange_disconnect(self._h))
            if self.world > 1:
                import torch.distributed as dist
                dist.barrier()
            self._L.tr_exchange_destroy(self._h)
            self._h = None
            if self.world > 1:
                dist.barrier()   # (... and nobody exports new slots at an address a peer is still unmapping)
