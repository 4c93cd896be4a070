"""ctypes loader for lib/libtiny_renderer.so (built in-tree from csrc/ by `make`)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = (os.environ.get("TR_LIBRARY") and os.path.abspath(os.environ["TR_LIBRARY"])) or os.path.join(_HERE, "lib", "libtiny_renderer.so")

TR_OK = 0
TR_E_INVALID = -1
TR_E_UNKNOWN_PIPELINE = -2
TR_E_BAD_POLYGON = -3
TR_E_SINGULAR = -4
TR_E_OOB_LOOKUP = -5
TR_E_HIP = -6
TR_E_IO = -7
TR_E_FORMAT = -8
TR_E_BIN_OVERFLOW = -9
TR_E_NOMEM = -10
TR_E_EXCHANGE = -11
TR_E_RCCL = -12
TR_EXCHANGE_PEER = 0
TR_EXCHANGE_RCCL = 1
TR_EXCHANGE_HANDLE_BYTES = 256

TR_OPT_WINNER_TAP = 0x1
TR_OPT_TILE_STAMPS = 0x2
TR_OPT_NO_AUTO_GROUP = 0x4
TR_OPT_TRUST_FRAME_BUFFERS = 0x8
TR_OPT_STORE_DEPTH = 0x10


class TinyRendererError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tiny_renderer error %d: %s" % (code, message))
        self.code = code


class Mesh(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_float)), ("tex", C.POINTER(C.c_float)),
                ("nrm", C.POINTER(C.c_float)), ("idx", C.POINTER(C.c_uint32)),
                ("n_pos", C.c_uint32), ("n_tex", C.c_uint32), ("n_nrm", C.c_uint32),
                ("n_tri", C.c_uint32)]


class ImageRgb8(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("w", C.c_uint32), ("h", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("band_row0", C.c_uint32), ("band_row1", C.c_uint32), ("stream", C.c_void_p),
                ("frame_buffer_device", C.c_void_p), ("bin_capacity", C.c_uint64),
                ("tile_waves", C.c_uint32), ("tile_mode", C.c_uint32), ("frames_per_launch", C.c_uint32),
                ("max_frame_slots", C.c_uint32)]


class FrameParams(C.Structure):
    _fields_ = [("light", C.c_float * 3), ("look_from", C.c_float * 3), ("look_at", C.c_float * 3),
                ("up", C.c_float * 3)]


class BandTiles(C.Structure):
    """tr_band_tiles (include/tiny_renderer.h): a scene's band of a frame buffer, tile by tile."""
    _fields_ = [("frame_buffer_device", C.c_void_p), ("clean_device", C.c_void_p), ("width", C.c_uint32),
                ("height", C.c_uint32), ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("first_tile_row", C.c_int32),
                ("band_y0", C.c_int32), ("band_y1", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double), ("frames", C.c_uint64)]


class Uniforms(C.Structure):
    _fields_ = [("camera_direction", C.c_float * 3), ("t_light_direction", C.c_float * 3),
                ("vpmv", C.c_float * 16), ("i_vpmv", C.c_float * 16), ("m", C.c_float * 16),
                ("i_m", C.c_float * 16), ("it_m", C.c_float * 16),
                ("shadow_matrix", C.c_float * 16)]


# Every symbol include/tiny_renderer.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
SYMBOLS = {
    "tr_scene_create": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(Mesh), C.POINTER(ImageRgb8),
                                  C.c_char_p, C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "tr_scene_destroy": (None, [C.c_void_p]),
    "tr_scene_clear": (C.c_int, [C.c_void_p]),
    "tr_scene_set_light_direction": (C.c_int, [C.c_void_p, _FP]),
    "tr_scene_set_camera": (C.c_int, [C.c_void_p, _FP, _FP, _FP]),
    "tr_scene_render": (C.c_int, [C.c_void_p]),
    "tr_scene_set_auto_group": (C.c_int, [C.c_void_p, C.c_int]),
    "tr_scene_render_frames": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "tr_scene_frames_per_launch": (C.c_int, [C.c_void_p]),
    "tr_scene_frames_kept": (C.c_int, [C.c_void_p]),
    "tr_scene_select_frame": (C.c_int, [C.c_void_p, C.c_uint32]),
    "tr_scene_get_frame_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_get_frame_buffer_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_host_alloc": (C.c_void_p, [C.c_size_t]),
    "tr_host_free": (None, [C.c_void_p]),
    "tr_scene_host_buffer_written": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_get_z_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_get_shadow_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_read_z_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_read_shadow_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_read_winner_u32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_sync": (C.c_int, [C.c_void_p]),
    "tr_scene_flush": (C.c_int, [C.c_void_p]),
    "tr_scene_frame_buffer_device": (C.c_void_p, [C.c_void_p]),
    "tr_scene_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_scene_set_frame_buffer_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_band_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tr_scene_debug_tile_stamps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "tr_scene_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tr_scene_profile_read": (C.c_int, [C.c_void_p, C.POINTER(KernelTime), C.c_int]),
    "tr_scene_profile_frame_intervals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "tr_selftest_device_math": (C.c_int, [C.c_int, _FP, _FP, C.c_uint32] + [C.c_void_p] * 5),
    "tr_selftest_shadow_fetch": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
                                 + [C.c_void_p] * 6),
    "tr_exchange_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tr_exchange_create_backend": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "tr_exchange_bytes_sent": (C.c_uint64, [C.c_void_p]),
    "tr_exchange_frame": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "tr_exchange_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_exchange_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tr_exchange_all_gather": (C.c_int, [C.c_void_p, C.c_uint32, C.c_size_t, C.c_size_t, C.c_void_p]),
    "tr_exchange_set_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tr_exchange_all_gather_tiles": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(BandTiles), C.c_void_p]),
    "tr_scene_band_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(BandTiles)]),
    "tr_exchange_status": (C.c_int, [C.c_void_p]),
    "tr_exchange_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "tr_exchange_disconnect": (C.c_int, [C.c_void_p]),
    "tr_exchange_destroy": (None, [C.c_void_p]),
    "tr_selftest_device_unary": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "tr_pipeline_count": (C.c_int, []),
    "tr_pipeline_name": (C.c_char_p, [C.c_int]),
    "tr_prepare_uniforms": (C.c_int, [C.c_int, C.POINTER(Uniforms), C.c_uint32, C.c_uint32,
                                      _FP, _FP, _FP, _FP]),
    "tr_load_obj": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(Mesh))]),
    "tr_free_mesh": (None, [C.POINTER(Mesh)]),
    "tr_load_tga_rgb8": (C.c_int, [C.c_char_p, C.POINTER(ImageRgb8)]),
    "tr_free_image": (None, [C.POINTER(ImageRgb8)]),
    "tr_save_tga_rgb8": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "tr_save_png_rgb8": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "tr_last_error": (C.c_char_p, []),
    "tr_abi_version": (C.c_int, []),
    "tr_specular_exact": (C.c_int, []),
}


def library_path():
    return _LIB


def build_library(force=False, quiet=True):
    """Compile csrc/ for gfx950 with hipcc (csrc/Makefile) into lib/libtiny_renderer.so."""
    cmd = ["make", "-C", _CSRC, "-j4"] + (["-B"] if force else [])
    if quiet:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return _LIB


_lib = None


def load_library():
    """Loads the C-ABI library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise TinyRendererError(TR_E_HIP, "%s is missing: run `make -C %s` (or "
                                    "__graft_entry__.build()); there is no CPU fallback" % (_LIB, _CSRC))
        L = C.CDLL(_LIB)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("TR_LIBRARY"):   # an older build loaded for an A/B measurement: it lacks newer entry points
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code < 0:
        msg = load_library().tr_last_error()
        raise TinyRendererError(code, msg.decode() if msg else "")
    return code
