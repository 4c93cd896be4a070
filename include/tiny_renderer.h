/*
 * tiny_renderer.h -- C ABI of the MI355X-native triangle-fill path.
 *
 * Drop-in boundary: the reference has no FFI; its caller (src/app.rs:137-146,170,208-213)
 * talks to the Rust methods of `Scene` (src/scene.rs:44-269).  Each entry point below names the
 * method it replaces.  A Rust host binds these with an `extern "C"` block;
 * everything is plain pointers, sizes and int status codes -- no exceptions cross this line.
 *
 * Call protocol per frame, as in app.rs:170,208-213:
 *     tr_scene_clear -> tr_scene_set_light_direction -> tr_scene_set_camera ->
 *     tr_scene_render -> tr_scene_get_frame_buffer
 * `render` does not clear; calling it twice without `clear` depth-tests against the previous
 * result exactly like the reference.  A tr_scene is not thread-safe (the reference's Scene is
 * not even Send, shader.rs:85-87).  All rendering runs on the GPU; there is no CPU fallback:
 * tr_scene_create fails with TR_E_HIP when no gfx950 device is usable.
 */
#ifndef TINY_RENDERER_H
#define TINY_RENDERER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: what this header declares is all it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define TR_ABI_VERSION 3   /* 3: tr_options.max_frame_slots, TR_OPT_STORE_DEPTH, tr_exchange_set_ranges, 64 exchange slots */

/* Status codes.  0 = ok, negative = failure (the reference panics at the cited site). */
enum {
    TR_OK = 0,
    TR_E_INVALID = -1,          /* bad argument / NULL handle */
    TR_E_UNKNOWN_PIPELINE = -2, /* shader.rs:108 */
    TR_E_BAD_POLYGON = -3,      /* scene.rs:218, or an index outside positions/tex_coords/normals */
    TR_E_SINGULAR = -4,         /* try_inverse().unwrap(): shader.rs:224,277,278,631; :921 */
    TR_E_OOB_LOOKUP = -5,       /* device error word: texture / shadow-buffer index out of range
                                   (util.rs:40,52,68,82; shader.rs:778,912,935) or w == 0
                                   (shader.rs:158); the frame is produced but parity is undefined */
    TR_E_HIP = -6,              /* HIP runtime failure or no usable device */
    TR_E_IO = -7,               /* file missing / unreadable (app.rs:94,99: `?`) */
    TR_E_FORMAT = -8,           /* unsupported OBJ / TGA content */
    TR_E_BIN_OVERFLOW = -9,     /* the (polygon, screen tile) pairs of one pass exceeded the record pool AND the frame
                                   could not be rendered again behind the caller's back (see tr_scene_sync); the
                                   pool has been grown: render the frame again, or size it with
                                   tr_options.bin_capacity.  No single tile has a capacity. */
    TR_E_NOMEM = -10,
    TR_E_EXCHANGE = -11,        /* multi-GPU frame exchange: a peer's band did not arrive */
    TR_E_RCCL = -12             /* RCCL backend of the frame exchange: librccl missing or a collective failed */
};

/* obj::raw::RawObj as the path reads it (util.rs:25-31, shader.rs:136-147,363-367,
 * scene.rs:216-226).  Indices are zero based; a polygon contributes its first three
 * (position, tex_coord, normal) triples. */
typedef struct tr_mesh {
    const float *pos;    /* n_pos * 3  (obj-rs keeps a 4th w component; the path ignores it) */
    const float *tex;    /* n_tex * 3 */
    const float *nrm;    /* n_nrm * 3 */
    const uint32_t *idx; /* n_tri * 9 : p0,t0,n0, p1,t1,n1, p2,t2,n2 */
    uint32_t n_pos, n_tex, n_nrm, n_tri;
} tr_mesh;

/* image::RgbImage: tightly packed rgb8, row 0 = top of the picture. */
typedef struct tr_image_rgb8 {
    const uint8_t *rgb;
    uint32_t w, h;
} tr_image_rgb8;

#define TR_OPT_WINNER_TAP 0x1u /* keep a per-pixel winning-polygon index (parity tap) */
#define TR_OPT_TILE_STAMPS 0x2u /* diagnostic: record per-tile start/end clocks of the last pass */
#define TR_OPT_NO_AUTO_GROUP 0x4u /* tr_scene_render submits every frame on its own (see tr_scene_render) */
/* A caller's frame buffers (tr_options.frame_buffer_device, tr_scene_set_frame_buffer_device, tr_scene_render_frames)
 * are written by nobody but the scene while they are its targets -- apart from rows outside the scene's band.  The
 * scene then keeps, per buffer, which tiles already hold the cleared colour, and a cleared frame does not store the
 * zeros of an empty tile again (a caller that double-buffers frames for an exchange: most of the frame, every frame).
 * Without the flag nothing is remembered about a caller's buffer from one tr_scene_set_frame_buffer_device /
 * tr_scene_render_frames call to the next: whatever wrote it in between -- a post-process, a memset, an allocator
 * handing the address to another tensor -- a cleared render produces every pixel of its band. */
#define TR_OPT_TRUST_FRAME_BUFFERS 0x8u
/* Transient depth off.  By default the colour pass of a CLEARED frame resolves its depth on the chip and does not write
 * it to the z buffer: nothing reads the z buffer of such a frame -- the next cleared frame overwrites it unseen -- and the
 * first consumer that does want it (tr_scene_get_z_buffer / tr_scene_read_z_f32, or a tr_scene_render without a clear,
 * which depth-tests against it: scene.rs:151) gets it from a repeat of the pass for the depth alone, so that what
 * callers observe is unchanged.  With this flag every colour pass writes its depth, as the reference's does. */
#define TR_OPT_STORE_DEPTH 0x10u

typedef struct tr_options {
    uint32_t struct_size;      /* = sizeof(tr_options) */
    int32_t device;            /* HIP device ordinal; -1 = current device */
    uint32_t flags;            /* TR_OPT_* */
    /* Screen-band shard (multi-GPU): this scene renders only output-image rows
     * [band_row0, band_row1) (row 0 = top, as returned by get_frame_buffer).
     * 0,0 = the whole frame. */
    uint32_t band_row0, band_row1;
    void *stream;              /* hipStream_t to enqueue on; NULL = library-owned stream */
    void *frame_buffer_device; /* device pointer to 3*W*H bytes to render into (e.g. the
                                  all-gather buffer); NULL = library-owned */
    uint64_t bin_capacity;     /* records in a pass's pool = (polygon, 128x16 screen tile) pairs of one pass of one
                                  frame; every tile gets exactly the records it needs from it.  0 = automatic:
                                  twice an estimate from the frame size and the polygon count -- polygons with
                                  boxes of side s = sqrt(2 W H / n) meet (1 + s/128)(1 + s/16) tiles each, half
                                  of them face the viewer -- at least 65 536 and at most 16 Mi records.  A pass
                                  that needs more grows the pools and the frame is rendered again. */
    uint32_t tile_waves;       /* wavefronts per 128x16 screen tile: 4, 8, 16, or 0 = automatic (more
                                  while the tiles cannot fill the GPU, 4 from 4096x4096 up).
                                  Speed only: results do not depend on it. */
    uint32_t tile_mode;        /* how a tile's wavefronts divide its work: 1 = each owns a column of the tile and
                                  sees every polygon of the bin, 2 = each owns a share of the bin and sees the
                                  whole tile (depth resolve through LDS atomics), 0 = automatic.
                                  Speed only: results do not depend on it. */
    uint32_t frames_per_launch; /* tr_scene_render_frames: frames rendered by one launch of each kernel (1..32), 0 =
                                  automatic (by tile count: 4 at 4096x4096, 32 for small frames; a call of several
                                  groups uses up to three times that per launch, a call of sixteen groups or more
                                  grows to 32).  Speed only. */
    uint32_t max_frame_slots;  /* upper bound on the scene's frame slots -- complete sets of render targets (z,
                                  colour, shadow buffer: 7 to 11 bytes per pixel each), one per frame of a group in
                                  flight, hence also on the frames per launch: 1..32, 0 = automatic (as many as the
                                  largest group, up to 32 within 8 GiB; a device without room for that falls back to
                                  the usual group by itself).  For callers that keep many scenes on one GPU. */
} tr_options;

typedef struct tr_scene tr_scene;

/* Scene::new (scene.rs:47-88).  tex[] = texture, normal_map, normal_map_tangent, specular_map
 * (the order of Scene::new's arguments).  Inputs are copied (the reference moves them).
 * Pipeline names: shader.rs:100-109 (`true_normal`, README.md:18, is accepted as an alias of
 * `normal_map`). */
int tr_scene_create(uint32_t width, uint32_t height, const tr_mesh *mesh,
                    const tr_image_rgb8 tex[4], const char *pipeline_name,
                    const tr_options *opts, tr_scene **out);
void tr_scene_destroy(tr_scene *s);

int tr_scene_clear(tr_scene *s);                                  /* scene.rs:128-137 */
int tr_scene_set_light_direction(tr_scene *s, const float v[3]); /* scene.rs:140-142 */
int tr_scene_set_camera(tr_scene *s, const float look_from[3], const float look_at[3],
                        const float up[3]);                       /* scene.rs:145-149 */
/* scene.rs:151-268.  Asynchronous.  On the library's own stream a render that follows a clear may be held back
 * on the host until a few such frames have been issued (tr_scene_frames_per_launch) and is then rendered
 * together with them by fused kernel launches (as tr_scene_render_frames does; only the last frame's targets
 * are the scene's, which is all the per-frame protocol lets anybody see).  Every getter, tr_scene_sync,
 * tr_scene_flush and tr_scene_get_frame_buffer_async submit what is held back first, so a frame that is read
 * right after its render goes to the device alone, at once. */
int tr_scene_render(tr_scene *s);
int tr_scene_set_auto_group(tr_scene *s, int on); /* the same switch as TR_OPT_NO_AUTO_GROUP, at run time */

/* Many frames per call -- the throughput path (nothing of the kind upstream, whose caller renders one frame per
 * window refresh, app.rs:170-213).  Frame i of the call is exactly what
 *     tr_scene_clear; tr_scene_set_light_direction(frames[i].light);
 *     tr_scene_set_camera(frames[i].look_from, .look_at, .up); tr_scene_render
 * produces, but the frames of a group (tr_scene_frames_per_launch of them) are rendered TOGETHER, by one launch of
 * each kernel: a lone frame leaves the GPU draining for a third of its tile kernel at 4096x4096, and a small frame
 * never fills it.  Each frame of a group has render targets of its own ("frame slots": z, colour, shadow buffer),
 * handed out in rotation, so when the call returns its LAST tr_scene_frames_per_launch frames exist
 * (tr_scene_frames_kept).  The scene is left as the per-frame calls would leave it: light and camera of the last
 * frame, the last frame current for every getter and for a later tr_scene_render without clear.
 * frame_buffers_device: NULL (colour into the slots' own buffers) or n_frames device pointers of 3*W*H bytes each,
 * frame i's colour target (e.g. the all-gather buffers of a multi-GPU caller); a buffer belongs to the scene as in
 * tr_scene_set_frame_buffer_device.  Asynchronous like tr_scene_render: on a caller's stream all the frames'
 * kernels are enqueued when the call returns. */
typedef struct tr_frame_params {
    float light[3];
    float look_from[3], look_at[3], up[3];
} tr_frame_params;
int tr_scene_render_frames(tr_scene *s, uint32_t n_frames, const tr_frame_params *frames, void *const *frame_buffers_device);
int tr_scene_frames_per_launch(tr_scene *s); /* frames per group of this scene */
int tr_scene_frames_kept(tr_scene *s);       /* frames of the last tr_scene_render_frames call that still exist
                                                (0 after a tr_scene_render) */
/* Makes the frame `back` frames before the last one of that call (0 = the last) the scene's current frame: getters,
 * tr_scene_frame_buffer_device and later renders refer to its targets, light and camera. */
int tr_scene_select_frame(tr_scene *s, uint32_t back);

/* scene.rs:92-125.  Caller-owned host buffers of 3*W*H bytes, row 0 = top.  Synchronizes.
 * Returns the sticky device status (TR_E_OOB_LOOKUP, TR_E_BIN_OVERFLOW) of the frame. */
int tr_scene_get_frame_buffer(tr_scene *s, uint8_t *rgb);
int tr_scene_get_z_buffer(tr_scene *s, uint8_t *rgb);
int tr_scene_get_shadow_buffer(tr_scene *s, uint8_t *rgb);

/* Parity taps, not in the reference: raw buffers in the reference's internal layout
 * (index = x + y*W, row 0 = bottom), W*H elements. */
int tr_scene_read_z_f32(tr_scene *s, float *out);
int tr_scene_read_shadow_f32(tr_scene *s, float *out);
int tr_scene_read_winner_u32(tr_scene *s, uint32_t *out); /* needs TR_OPT_WINNER_TAP;
                                                             0xFFFFFFFF = no fragment */

/* Streaming frames out (the reference hands every frame to its window, app.rs:213-218): enqueue
 * the device-to-host copy of the frame behind the renders issued so far and return at once; `rgb`
 * (3*W*H bytes, row 0 = top) holds the frame after tr_scene_sync(); a later render is ordered after the copy.
 * Into memory from tr_host_alloc (page-locked, mapped into the device) only what has to travel does: the scene
 * knows which 128x16 tiles of the frame hold the cleared colour, and remembers per host buffer which tiles it
 * has written as zeros there -- those are skipped (widths that are multiples of 16; three quarters of a
 * 4096x4096 frame of the reference's model: 1.1 ms -> 0.3 ms per frame).  The buffer always ends up holding the
 * complete frame: the record of a buffer's zero tiles belongs to the scene that wrote it last and lapses when
 * anybody else writes the buffer -- another scene reading back into it (the library knows), or the caller, who
 * says so with tr_scene_host_buffer_written (no scene then assumes anything about its content); reading it needs
 * nothing.  A band scene (tr_options.band_row0/1) and any other host memory receive the whole frame buffer through
 * the copy engine. */
int tr_scene_get_frame_buffer_async(tr_scene *s, uint8_t *rgb);
void *tr_host_alloc(size_t bytes); /* page-locked host memory mapped into the device, NULL on failure */
void tr_host_free(void *p);
int tr_scene_host_buffer_written(tr_scene *s, void *p);

/* Device-resident access for callers that keep the frame on the GPU. */
int tr_scene_sync(tr_scene *s);                 /* wait for queued work; returns frame status */
int tr_scene_flush(tr_scene *s);                /* hand every render issued so far to the device (the library
                                                   may hold a few back to batch them); does not wait */
void *tr_scene_frame_buffer_device(tr_scene *s); /* 3*W*H bytes, row 0 = top */
int tr_scene_set_stream(tr_scene *s, void *hip_stream);
/* Swap the render target: renders issued after the call write `frame_buffer_device` (3*W*H bytes of
 * device memory, row 0 = top; NULL = the library's own buffer); renders already issued keep theirs.
 * A pending `clear` carries over (the next render produces every pixel of the new buffer);
 * without one the new buffer's content is taken as the frame so far.  For callers that double-buffer
 * the frame, e.g. to exchange frame f between GPUs while frame f+1 renders. */
int tr_scene_set_frame_buffer_device(tr_scene *s, void *frame_buffer_device);

/* Screen-band partition of a frame over the GPUs of a node (the reference's own clamp rectangle,
 * scene.rs:236-239, cut into row bands): rank r of n renders output rows [row0, row1), row 0 = top
 * -- the values for tr_options.band_row0/1.  Bands are disjoint, ordered by rank and cover the
 * frame; they are equal (what an in-place all-gather needs) exactly when n divides height. */
int tr_band_rows(uint32_t height, uint32_t n_ranks, uint32_t rank, uint32_t *row0, uint32_t *row1);

/* Multi-GPU frame exchange without RCCL (SURVEY.md 8e's hand-tuned alternative; nothing of the kind
 * upstream).  One process per GPU.  Each rank creates its end -- 1 to 64 full-size frame buffers
 * ("slots": pass their device pointers to tr_scene_create / tr_scene_set_frame_buffer_device) -- and
 * publishes a TR_EXCHANGE_HANDLE_BYTES record; once every rank has connected to all records (in rank
 * order; the host's own rendezvous carries them), tr_exchange_all_gather(slot, offset, bytes, stream)
 * pushes bytes [offset, offset + bytes) of the local slot into the same range of every peer's slot
 * with concurrent DMA-engine copies over xGMI, after the work queued on `stream` so far, and makes
 * `stream` wait until every peer's range has arrived in the local slot.  Collective: every rank calls it
 * for the same slots in the same order.  Failures of a peer surface as TR_E_EXCHANGE from
 * tr_exchange_status / tr_exchange_read after a device-side timeout (ten seconds; the environment
 * variable TR_EXCHANGE_TIMEOUT_MS, read at create, overrides it), never as a hang. */
#define TR_EXCHANGE_HANDLE_BYTES 256
typedef struct tr_exchange tr_exchange;
int tr_exchange_create(int device, uint32_t n_ranks, uint32_t rank, uint32_t n_slots, size_t frame_bytes, tr_exchange **out);
/* The same exchange with a choice of transport.  TR_EXCHANGE_PEER: the copies described above (tr_exchange_create).
 * TR_EXCHANGE_RCCL (north_star: "RCCL all-gather of the final framebuffer over xGMI"): connect builds
 * an RCCL communicator owned by the exchange (rank 0's record carries the ncclUniqueId: the host's rendezvous only
 * moves the 256-byte records, as for the peer transport; every rank must be inside tr_exchange_connect at the same
 * time) and tr_exchange_all_gather is ONE in-place ncclAllGather on `hip_stream`; the ranks' byte ranges must then be
 * equal pieces of one range, in rank order -- tr_band_rows with a height the ranks divide.  librccl is loaded when
 * such an exchange is created; failures are TR_E_RCCL.  A Rust (or C) host needs no torch for either. */
#define TR_EXCHANGE_PEER 0
#define TR_EXCHANGE_RCCL 1
int tr_exchange_create_backend(int device, uint32_t n_ranks, uint32_t rank, uint32_t n_slots, size_t frame_bytes, int backend,
                               tr_exchange **out);
uint64_t tr_exchange_bytes_sent(tr_exchange *x); /* bytes this rank has pushed to its peers since the exchange was created */
void *tr_exchange_frame(tr_exchange *x, uint32_t slot);
int tr_exchange_export(tr_exchange *x, void *record /* TR_EXCHANGE_HANDLE_BYTES */);
int tr_exchange_connect(tr_exchange *x, const void *records /* n_ranks * TR_EXCHANGE_HANDLE_BYTES */);
int tr_exchange_all_gather(tr_exchange *x, uint32_t slot, size_t offset, size_t bytes, void *hip_stream);
/* Declares every rank's byte range of a frame (n_ranks offsets and sizes, disjoint: the bands of tr_band_rows), the same
 * on all ranks, before the first tr_exchange_all_gather.  The peer transport's dense exchange then PULLS: a rank copies
 * its peers' ranges out of their mapped slots into its own slot, and nothing but 4-byte flags is ever written into
 * another rank's memory -- a peer that is late or gone costs this rank its own frame (TR_E_EXCHANGE), never a slot the
 * owner had not opened (the push form's copy engines cannot be predicated on the error word).  tr_exchange_all_gather
 * must then be called with this rank's declared range.  NULL, NULL: back to the push form.  RCCL transport: checked, unused. */
int tr_exchange_set_ranges(tr_exchange *x, const size_t *offsets, const size_t *bytes);
/* The SPARSE form of the exchange: the band of `slot`'s frame that scene `s` renders goes to the peers tile by tile
 * (128 x 16 pixels), and a tile that holds the cleared colour here AND held it the last time this rank wrote the
 * peer's copy does not travel at all -- three quarters of a 4096x4096 frame of the reference's model.  The scene
 * knows which tiles those are (the fast-clear flags of its frame buffers: tr_scene_band_tiles); the exchange keeps
 * the record of what each peer's copy holds.  The slot must be the frame buffer the scene rendered into
 * (tr_scene_set_frame_buffer_device / tr_scene_render_frames) and nobody else may write this rank's band of the
 * peers' copies.  Same ordering and collective rules as tr_exchange_all_gather; a tr_exchange_all_gather on the slot
 * in between is allowed (it resets the record).  TR_EXCHANGE_PEER: one kernel per peer storing through the mapped
 * slots over xGMI.  TR_EXCHANGE_RCCL (a collective's sizes are fixed by the host before the frame's coverage is
 * known) and frames whose width is not a multiple of 16: the dense exchange of the band's rows.
 * tr_exchange_bytes_sent counts what really travelled (it waits for the device). */
typedef struct tr_band_tiles {
    const void *frame_buffer_device; /* the frame buffer the flags describe */
    const uint32_t *clean_device;    /* tiles_x * tiles_y flags on the device, row-major: != 0 = the tile's pixels are zeros */
    uint32_t width, height;          /* the frame */
    uint32_t tiles_x, tiles_y;       /* tile grid of the band; tiles are 128 x 16 pixels */
    int32_t first_tile_row;          /* tile row (y up, rows of 16 pixels from the bottom) of grid row 0 */
    int32_t band_y0, band_y1;        /* scene rows [y0, y1), y up, that the scene renders; buffer row = height - 1 - y */
} tr_band_tiles;
/* Queues what is still held back of the scene's frames and describes the tiles of `frame_buffer_device` (one of the
 * buffers the scene has rendered into; NULL = the current one).  The flags are read by work queued AFTER the frame
 * on the scene's stream or on a stream that waits for it. */
int tr_scene_band_tiles(tr_scene *s, const void *frame_buffer_device, tr_band_tiles *out);
int tr_exchange_all_gather_tiles(tr_exchange *x, uint32_t slot, const tr_band_tiles *tiles, void *hip_stream);
int tr_exchange_status(tr_exchange *x);
int tr_exchange_read(tr_exchange *x, uint32_t slot, void *host, size_t bytes); /* waits for the device, copies a slot out */
/* Tearing a peer exchange down takes two steps when its ranks go on living (they create another exchange, say): a rank
 * must not FREE slots a peer still has mapped (what HIP leaves undefined for exported memory), so every rank first
 * unmaps its peers' slots and flags -- tr_exchange_disconnect: waits for the device, after which the exchange can only
 * be destroyed --, the host's rendezvous confirms that all have (a barrier), and then each destroys its end.
 * tr_exchange_destroy alone does both at once: fine when the process ends anyway. */
int tr_exchange_disconnect(tr_exchange *x);
void tr_exchange_destroy(tr_exchange *x);

/* Diagnostic (TR_OPT_TILE_STAMPS): for each tile of the last colour pass {start, end} in 100 MHz
 * ticks, polygons in its bin, hardware id, {bin staged, coverage done} ticks, 2 spare.  `out`
 * holds 8 * n_tiles entries; returns n_tiles. */
int tr_scene_debug_tile_stamps(tr_scene *s, uint64_t *out, uint32_t cap_tiles);

/* Per-kernel device timing with HIP events on the scene's stream (bench roofline leg). */
typedef struct tr_kernel_time {
    char name[32];
    uint64_t launches;
    double total_ms;
    uint64_t frames; /* frames those launches covered (= launches, except for tr_scene_render_frames' fused launches) */
} tr_kernel_time;
int tr_scene_profile_enable(tr_scene *s, int on);
/* Fills up to `cap` entries, returns the number of kernels or a negative status. */
int tr_scene_profile_read(tr_scene *s, tr_kernel_time *out, int cap);
/* Frame times of the profiled renders: microseconds between the completions of consecutive frames'
 * (colour-pass) tile kernels, in issue order.  Returns how many were written (<= cap). */
int tr_scene_profile_frame_intervals(tr_scene *s, float *out_us, int cap);

/* Device self-test of the arithmetic primitives the kernels substitute for the reference's:
 * Rust `as` casts (f32 -> u32 / i32 / u8) and x / d through a shared reciprocal.  Inputs and
 * outputs are host arrays of n elements; div_ref receives the device's plain x / d. */
int tr_selftest_device_math(int device, const float *x, const float *d, uint32_t n, uint32_t *out_u32,
                            int32_t *out_i32, uint32_t *out_u8, float *out_div, float *out_div_ref);

/* Device self-test of the shadow-buffer lookup of the shadow / occlusion closures (shader.rs:774-778, 909-912,
 * 932-935: (round(x) as u32 + round(y) as u32 * width) as usize, wrapping) as the kernels perform it THROUGH the
 * shadow buffer's per-tile fast-clear flags: `stale` is a width x height buffer whose tiles with a non-zero flag in
 * `sclean` (one word per 128 x 16 tile, row-major) hold arbitrary values, `plain` the same buffer with those tiles
 * written as f32::MIN.  For each of the n coordinates the value bits and error bits (4 = index out of range) of the
 * plain lookup in `plain` and of the flagged lookup in `stale` are returned; they must be equal -- also for a
 * column beyond the row and for a row k * 2^32 / width + r, which wraps around 2^32 back into the buffer. */
int tr_selftest_shadow_fetch(int device, uint32_t width, uint32_t height, const float *plain, const float *stale,
                             const uint32_t *sclean, uint32_t n, const float *x, const float *y, uint32_t *out_plain,
                             uint32_t *out_flagged, uint32_t *err_plain, uint32_t *err_flagged);

/* Exhaustive device check of the kernels' own correctly rounded reciprocal (which = 0) and square
 * root (which = 1) for pixel pairs (csrc/tr_pk.h rcp2 / sqrt2: hardware estimate + fused residual
 * corrections) against the compiler's IEEE `1.0f / x` and `sqrtf`: every f32 with binary exponent in
 * [exp_lo, exp_hi].  which = 2: the colour channels' cast-and-insert (v_cvt_pk_u8_f32) against the Rust `as u8` it
 * stands for, x and -x; exp_lo = -127 .. exp_hi = 128 covers every f32 (zeros, subnormals, infinities, NaNs).
 * n_bad counts differing results, bad_bits receives up to 16 of the arguments. */
int tr_selftest_device_unary(int device, int which, int exp_lo, int exp_hi, uint64_t *n_tested, uint64_t *n_bad,
                             uint32_t bad_bits[16]);

/* 1 when this build's specular pipeline (shader.rs:472-543, the only one that calls powf) returns
 * the host C library's powf bit for bit -- the library was built against a glibc whose powf
 * tables it could read (csrc/gen_powf_tables.py); 0: the device library's powf, within 1 ulp. */
int tr_specular_exact(void);

/* shader.rs:97-112 registry */
int tr_pipeline_count(void);
const char *tr_pipeline_name(int i);

/* shader.rs:183-279 prepares, host-side (no GPU needed).  kind: 0 default_prepare,
 * 1 shadow_pass_prepare_1, 2 shadow_pass_prepare_2.  Matrices column-major (nalgebra). */
typedef struct tr_uniforms {
    float camera_direction[3];
    float t_light_direction[3];
    float vpmv[16];
    float i_vpmv[16];
    float m[16];
    float i_m[16];
    float it_m[16];
    float shadow_matrix[16];
} tr_uniforms;
int tr_prepare_uniforms(int kind, tr_uniforms *u, uint32_t width, uint32_t height,
                        const float light[3], const float look_from[3],
                        const float look_at[3], const float up[3]);

/* Asset loading (app.rs:87-131): obj-rs `parse_obj` and image `open(..).into_rgb8()`
 * counterparts.  Returned objects are owned by the library; free with the matching call. */
int tr_load_obj(const char *path, tr_mesh **out);
void tr_free_mesh(tr_mesh *m);
int tr_load_tga_rgb8(const char *path, tr_image_rgb8 *out);
void tr_free_image(tr_image_rgb8 *img);
/* Frame writer (no counterpart upstream: the reference shows frames in a window): uncompressed
 * 24-bit TGA, top-left origin, i.e. exactly what tr_scene_get_frame_buffer returns. */
int tr_save_tga_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h);
/* The same frame as an 8-bit RGB PNG (stored deflate blocks: no compression library needed). */
int tr_save_png_rgb8(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h);

const char *tr_last_error(void);
int tr_abi_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif
