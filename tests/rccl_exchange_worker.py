"""One rank of the library's RCCL exchange (tr_exchange_create_backend(..., TR_EXCHANGE_RCCL)), driven through
the C ABI alone -- no torch, as a Rust host would: a band scene renders into the exchange's frame slot on a stream
the library owns, the in-place ncclAllGather of the band runs on that stream, and the slot holds the oracle's
frame.  With one rank the gather moves nothing, but the communicator is built from the published record, the
byte-range rules are checked and the collective is really issued.  Started by tests/test_gpu_parity.py."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_renderer_amd as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tiny_renderer_amd import _lib as TL  # noqa: E402

L = T.load_library()
W, Hh = 512, 384
mesh, texs = T.synthetic_scene(n_lat=12, n_lon=24, tex_size=256)
h = C.c_void_p()
TL.check(L.tr_exchange_create_backend(0, 1, 0, 2, W * Hh * 3, TL.TR_EXCHANGE_RCCL, C.byref(h)))
rec = C.create_string_buffer(TL.TR_EXCHANGE_HANDLE_BYTES)
TL.check(L.tr_exchange_export(h, rec))
TL.check(L.tr_exchange_connect(h, rec.raw))
band = T.band_rows(Hh, 1, 0)
slot = L.tr_exchange_frame(h, 0)
gpu = T.Scene(W, Hh, mesh, texs, "phong", frame_buffer_device=slot, band_rows=band, trust_frame_buffers=True)
cpu = O.Scene(W, Hh, mesh, texs, "phong")
for ca in (0.0, 0.8, 1.9):
    for s in (gpu, cpu):
        s.clear(), s.set_light_direction(H.light(0.4)), s.set_camera(*H.camera(ca)), s.render()
    assert gpu.sync() == 0
    TL.check(L.tr_exchange_all_gather(h, 0, 0, W * Hh * 3, None))
    got = np.empty((Hh, W, 3), np.uint8)
    TL.check(L.tr_exchange_read(h, 0, got.ctypes.data, got.nbytes))
    assert np.array_equal(got, cpu.get_frame_buffer()), "angle %.1f" % ca
# the sparse call on the RCCL transport is answered with the dense all-gather of the band's rows (a collective's sizes
# are fixed before the frame's coverage is known): same frame, and tiles that describe another buffer are refused
for s in (gpu, cpu):
    s.clear(), s.set_light_direction(H.light(-0.3)), s.set_camera(*H.camera(2.6)), s.render()
tiles = gpu.band_tiles(slot)
assert tiles.width == W and tiles.height == Hh and tiles.tiles_x == (W + 127) // 128 and tiles.band_y0 == 0 and tiles.band_y1 == Hh
TL.check(L.tr_exchange_all_gather_tiles(h, 0, C.byref(tiles), None))
TL.check(L.tr_exchange_read(h, 0, got.ctypes.data, got.nbytes))
assert np.array_equal(got, cpu.get_frame_buffer()), "sparse call, RCCL transport"
other = TL.BandTiles.from_buffer_copy(tiles)
other.frame_buffer_device = (other.frame_buffer_device or 0) + 4096
assert L.tr_exchange_all_gather_tiles(h, 0, C.byref(other), None) == TL.TR_E_INVALID
assert L.tr_scene_band_tiles(gpu._h, C.c_void_p(12345), C.byref(other)) == TL.TR_E_INVALID   # a buffer the scene never rendered into
# ranges that are not equal pieces in rank order are refused (rank 0 of 1: any offset > 0 leaves no room)
assert L.tr_exchange_all_gather(h, 0, 16, W * Hh * 3, None) == TL.TR_E_INVALID
assert L.tr_exchange_status(h) == 0
assert L.tr_exchange_bytes_sent(h) == 0      # one rank pushes nothing
gpu.close()
L.tr_exchange_destroy(h)
print("OK")
