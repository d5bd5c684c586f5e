/*
 * mbk.h -- C ABI of libmbk_hip.so, the MI355X (gfx950) Mandelbrot tile kernel library.
 *
 * The reference (ofsouzap/DistributedMandelbrot) has NO in-process plugin / FFI interface: its only
 * compute path is a numba-CUDA ufunc inside a Python worker script.  The seam a drop-in must honour
 * is the worker<->Distributer TCP protocol (spoken by distributedmandelbrot_amd/worker.py); the
 * in-process seam is the worker's `process_workload(level, mrd, index_real, index_imag)`.  Each entry
 * point below names the reference code it replaces.  Paths are relative to the reference root;
 * "WorkerCUDA.py" = DistributedMandelbrotWorkerCUDA/DistributedMandelbrotWorkerCUDA.py.
 *
 * Conventions: plain C symbols; every call returns an int status (MBK_OK == 0); no exceptions, no
 * C++ or torch types cross the boundary; output buffers are caller-owned; one mbk_ctx per GPU; a
 * ctx is NOT thread-safe (use one host thread per ctx -- ctypes releases the GIL during calls).
 * There is NO CPU fallback: without a usable gfx950 device mbk_create fails with MBK_ERR_NO_DEVICE.
 *
 * Arithmetic contract: IEEE-754 binary64, every operation individually
 * rounded, no FMA contraction of the reference's expressions; iteration counts are bit-identical to
 * a strict evaluation of WorkerCUDA.py:39-68 on coordinates bit-identical to np.linspace's.
 */
#ifndef MBK_H
#define MBK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBK_ABI_VERSION 4

/* DataChunk.cs:20 (dataChunkRange), WorkerCUDA.py:80 (definition = 4096). */
#define MBK_CHUNK_DEFINITION 4096u
/* DataChunk.cs:27 (dataChunkSize): bytes in one tile result on the wire (Distributer.cs:415-416). */
#define MBK_CHUNK_BYTES (4096u * 4096u)

enum mbk_status {
    MBK_OK = 0,
    MBK_ERR_INVALID = 1,   /* bad argument (NULL pointer, empty window outside the view, mrd > INT32_MAX ...) */
    MBK_ERR_NO_DEVICE = 2, /* no HIP device / device index out of range / not a gfx950-class GPU */
    MBK_ERR_HIP = 3,       /* a HIP runtime call failed; see mbk_last_error */
    MBK_ERR_NOMEM = 4,
    MBK_ERR_NET = 5        /* mbk_worker_run / mbk_feeder_run: a socket call failed or the server spoke out of protocol */
};

/* flags for the compute calls */
#define MBK_WANT_COUNTS 0x1u /* write int32 escape indices (what calc_mb_value returns, WorkerCUDA.py:39) */
#define MBK_WANT_BYTES 0x2u  /* write the quantised uint8 (WorkerCUDA.py:96-98) -- fused on device */
/* Kernel selection (bits 8..11).  0 = default: "scan" or "group", decided per launch from a 256-pixel host probe of
 * the window (MBK_OPT_HEAVY_SHARE) -- a deterministic function of the window, not of earlier launches.  The others
 * exist so that the parity tests and bench.py can A/B every shipped kernel variant; all of them are bit-exact. */
#define MBK_KERNEL_SHIFT 8
#define MBK_KERNEL_MASK 0xF00u
#define MBK_KERNEL_DEFAULT 0x000u
#define MBK_KERNEL_SIMPLE 0x100u /* one lane per pixel, compiler-scheduled loop, literal (2*zr)*zi form */
#define MBK_KERNEL_ASM 0x200u    /* one lane per pixel, hand-scheduled gfx950 loop */
#define MBK_KERNEL_REFILL 0x300u /* persistent waves with lane refill (deep-zoom divergence) */
#define MBK_KERNEL_GROUP 0x400u  /* hand-scheduled loop, bailout tested once per 16 (interior) / 8 steps + exact replay; one workgroup per 8x8 block */
#define MBK_KERNEL_SCAN 0x500u   /* a persistent light pass that finishes every block whose pixels escape within 4 steps,
                                    then one workgroup (the "group" code) per block it listed as unfinished */

/* Arithmetic of the escape loop (bit 12).  Default = IEEE binary64, the reference's arithmetic.
 * MBK_PRECISION_F32 is BASELINE config 4's "fp32 kernel variant" -- NOT in the reference (its only
 * signature is int32(float64, float64, int32), WorkerCUDA.py:39): coordinates are generated in fp64
 * exactly as above, rounded once to binary32, and the loop runs in strict (contraction-free) binary32.
 * Its oracle is oracle/mandel_oracle.c:mbo_escape_f32.  Supported by the asm / group kernels. */
#define MBK_PRECISION_F32 0x1000u

/* Host-buffer calls with MBK_WANT_BYTES (bit 13): copy the quantised bytes to the host only if the tile is not
 * uniform.  The stats reduction runs on the device anyway; when it reports all_bytes_zero ("Never",
 * DataChunk.cs:82) or all_bytes_one ("Immediate", :87) the 16 MiB copy is skipped and h_bytes is left
 * UNTOUCHED -- the caller takes the constant from mbk_stats.  3 tiles in 4 of a pyramid level are of that
 * kind, and the copy (0.31 ms pinned) is 5x their kernel time.  Honoured by mbk_view_compute,
 * mbk_view_submit and mbk_datachunk_submit_ex; the decision is made in mbk_wait. */
#define MBK_LAZY_UNIFORM 0x2000u

typedef struct mbk_ctx mbk_ctx;

/*
 * A view: width x height samples of [start_r, start_r+range_r] x [start_i, start_i+range_i], both
 * endpoints included -- exactly gen_arrays' two np.linspace calls (WorkerCUDA.py:24-32) with
 * `definition` generalised to width/height.  The window (col0,row0,ncols,nrows) selects which
 * samples are computed (row bands are the multi-GPU shard unit); coordinates always come from the
 * FULL view's linspace, so a banded view is bit-identical to the whole one.
 * Output layout: element (row-row0)*ncols + (col-col0); real is the fast axis (np.tile, :34),
 * imaginary the slow one (np.repeat, :35); row 0 is start_i.
 */
typedef struct mbk_view {
    double start_r, start_i;
    double range_r, range_i;
    uint32_t width, height;
    uint32_t col0, row0, ncols, nrows;
} mbk_view;

typedef struct mbk_stats {
    float kernel_ms;           /* hipEvent time of the escape-time kernel launch(es) on the slot's stream.  With
                                  MBK_OPT_PREPASS_OVERLAP = 1 (the default) the dispatch-order pre-pass (~13 us: a fill +
                                  classify_blocks_kernel) runs on an auxiliary stream beside the previous tile and is NOT
                                  inside this interval unless the tile kernel had to wait for it; with 0 it is */
    float d2h_ms;              /* hipEvent time of the device->host copies */
    uint64_t pixel_iterations; /* sum over pixels of (count if count>0 else mrd-1), from the kernel's own counts: the
                                  REFERENCE's iterations for this output (with the cycle test on, fewer are executed) */
    uint64_t never_pixels;     /* pixels with count == 0 (never escaped) */
    uint32_t all_bytes_zero;   /* 1 iff every quantised byte == 0: DataChunk.IsNeverChunk,     DataChunk.cs:82 */
    uint32_t all_bytes_one;    /* 1 iff every quantised byte == 1: DataChunk.IsImmediateChunk, DataChunk.cs:87 */
    uint64_t rle_runs;         /* runs of equal bytes in the quantised tile: the RLE codec (DataChunkSerializer.cs:56-100)
                                  writes 5 bytes per run, so its size is 1 + 5*rle_runs vs 1 + n for Raw (0 if no bytes) */
} mbk_stats;

typedef struct mbk_device_info {
    char name[128];
    char arch[64];
    int compute_units;
    int clock_mhz;       /* max engine clock */
    int wavefront_size;
    uint64_t total_mem;
} mbk_device_info;

int mbk_abi_version(void);

/* Number of HIP devices visible to this process. */
int mbk_device_count(int *count);

/* One context per GPU: device selection, a private stream, events, grow-on-demand device buffers.
 * Replaces the implicit default-CUDA-device context numba creates at WorkerCUDA.py:87. */
int mbk_create(int device, mbk_ctx **out);
void mbk_destroy(mbk_ctx *ctx);

/* Message for the last failing call on this ctx (ctx may be NULL for mbk_create failures).
 * The string stays valid until the next call on the same ctx / thread. */
const char *mbk_last_error(const mbk_ctx *ctx);

int mbk_get_device_info(mbk_ctx *ctx, mbk_device_info *info);
/* PCI bus id of the ctx's GPU as "dddd:bb:dd.f" (hipDeviceGetPCIBusId); len >= 16.  What a multi-GPU job reports
 * per rank so that a reader can tell N distinct GPUs took part (there is no RCCL communicator to ask). */
int mbk_device_pci_bus_id(mbk_ctx *ctx, char *buf, int len);

/* Pinned host memory for result buffers (direct DMA target of the D2H copy).  Optional: any host
 * pointer is accepted by the compute calls. */
int mbk_host_alloc(mbk_ctx *ctx, uint64_t bytes, void **out);
int mbk_host_free(mbk_ctx *ctx, void *ptr);

/* Tile geometry: WorkerCUDA.py:75-78 == DataChunk.cs:32-33,59-66.
 * range = 4/level; start = -2 + range*index (each operation individually rounded).
 * MBK_ERR_INVALID if level == 0 or an index >= level (DataChunk.cs:99-106). */
int mbk_datachunk_geometry(uint32_t level, uint32_t index_real, uint32_t index_imag,
                           double *start_r, double *start_i, double *range);

/*
 * Asynchronous launch on DEVICE pointers, on the caller's HIP stream (a hipStream_t; NULL = HIP's null
 * stream, which is what PyTorch's default stream is -- NOT the ctx's private stream).
 * Replaces gen_arrays (WorkerCUDA.py:19-37: coordinates are generated in-kernel), the two H2D copies
 * (:87-88), the ufunc launch `calc_mb_value(r_device, i_device, mrd, out=out_device)` (:92) and,
 * with MBK_WANT_BYTES, the host quantiser (:96-98).  d_counts: int32[nrows*ncols] (or NULL without
 * MBK_WANT_COUNTS); d_bytes: uint8[nrows*ncols] (or NULL without MBK_WANT_BYTES).
 * mrd is the reference's "maximum recursion depth": at most mrd-1 updates, result in {0} U [1, mrd-1].
 * A launch may enqueue helper kernels (scan pass, dispatch-order pre-pass, work-queue reset) that use
 * scratch memory the ctx keeps PER STREAM (a few bytes per 8x8 block of the largest window seen on that
 * stream): launches on one stream are ordered, so any number may be queued, on any number of streams.
 */
int mbk_view_launch(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                    int32_t *d_counts, uint8_t *d_bytes, void *hip_stream);

/* Synchronous: launch + D2H into HOST buffers (either may be NULL according to flags) + stats.
 * Replaces WorkerCUDA.py:82-98 for a generic view. */
int mbk_view_compute(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                     int32_t *h_counts, uint8_t *h_bytes, mbk_stats *stats);

/* Synchronous: one 4096x4096 DataChunk tile, exactly the reference's
 * `process_workload(level, mrd, index_real, index_imag) -> uint8[16777216]` (WorkerCUDA.py:70-100).
 * h_bytes: MBK_CHUNK_BYTES bytes, the payload the worker sends after 0x20 (WorkerCUDA.py:168).
 * h_counts: optional int32[16777216] (NULL to skip its D2H). */
int mbk_datachunk(mbk_ctx *ctx, uint32_t level, uint32_t mrd, uint32_t index_real,
                  uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts, mbk_stats *stats);

/*
 * BASELINE config 5 -- continuous ("smooth") escape-time colouring.  NOT in the reference (its output
 * is the integer index quantised to a byte, WorkerCUDA.py:96-98); defined here as
 *     nu = n + 1 - log2(0.5 * ln |z_n|^2)   for a pixel that escapes at step n (|z_n|^2 >= 4 is the
 *                                            reference's own bailout value), and 0 if it never escapes,
 * in binary64.  n is the bit-exact count of the parity kernels (also returned); the logarithms are the
 * device's (tests allow 1e-12 against libm).  Asynchronous form on DEVICE pointers / caller's stream,
 * and synchronous form into HOST buffers.  d_counts / h_counts may be NULL.
 */
int mbk_view_launch_smooth(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                           int32_t *d_counts, double *d_smooth, void *hip_stream);
int mbk_view_compute_smooth(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                            int32_t *h_counts, double *h_smooth, mbk_stats *stats);

/* NOTE on slot 0: the synchronous calls (mbk_view_compute, mbk_datachunk, mbk_view_compute_smooth,
 * mbk_quantise_counts) and mbk_serialize_last work on slot 0's buffers; while a tile submitted on slot 0 has not
 * been waited for they return MBK_ERR_INVALID instead of touching them. */
/*
 * Several tiles in flight per context (each slot has its own HIP stream, events and device buffers, allocated on first
 * use: the D2H of one tile overlaps the kernels of the others, and the host's per-tile work -- enqueue, wake-up -- hides
 * behind the GPU's).  mbk_datachunk_submit enqueues kernel + stats reduction + D2H for a tile on `slot`
 * (0 .. MBK_SLOTS-1) and returns at once; mbk_wait blocks until that slot's tile is in h_bytes / h_counts (use pinned
 * memory, mbk_host_alloc, or the copy is not asynchronous) and fills stats.  A slot holds one tile at a time.  Slot 0 is
 * also what the synchronous calls use.  Same single-host-thread rule as everything else on a ctx.  Two slots were
 * round 1-4's pipeline (1 973 tiles/s on level 16 against a bound of 2 490: every wait exposed the host's enqueue +
 * wake-up latency); four keep the copy engine and the CUs fed (profiles/r05).
 *
 * MBK_LAZY_UNIFORM (flags of mbk_datachunk_submit_ex / mbk_view_submit): the caller does not need h_bytes when the tile
 * turns out uniform (stats.all_bytes_zero / all_bytes_one: DataChunk.cs:82,87 "Never" / "Immediate" chunks, which the
 * server stores without a payload).  Then (1) a window that lies wholly outside |c| = 2 (with a margin of 1e-8) is
 * answered on the host without any GPU work when mrd >= 256: every count is 1, every byte 1 (proof:
 * tests/test_oracle.py) -- the corners of [-2, 2]^2 outside the inscribed circle, 1 - pi/4 = 21 % of the tiles of a deep
 * pyramid level (32 of level 16's 256; its other 160 "Immediate" tiles hold counts 1..4 and are computed); (2) the copy of a tile whose host probe says "all gone within 4 steps" is decided when the
 * statistics arrive; (3) every other tile is copied as usual.  h_bytes of a tile reported uniform is unspecified.
 */
#define MBK_SLOTS 4
/* Tiles the worker loops (mbk_worker_run, worker.run_pipelined) keep in flight on one ctx: the measured-best depth of the
 * host-buffer pipeline (profiles/r05/level16.log: level 16 at 2 / 3 / 4 in flight = 2 235 / 2 326 / 2 262 tiles/s, with
 * MBK_LAZY_UNIFORM 4 400 / 4 603 / 4 331); MBK_SLOTS is the capacity a caller of mbk_*_submit may use. */
#define MBK_WORKER_DEPTH 3
int mbk_datachunk_submit(mbk_ctx *ctx, int slot, uint32_t level, uint32_t mrd, uint32_t index_real,
                         uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts);
int mbk_datachunk_submit_ex(mbk_ctx *ctx, int slot, uint32_t level, uint32_t mrd, uint32_t index_real,
                            uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts, uint32_t flags /* MBK_LAZY_UNIFORM */);
int mbk_wait(mbk_ctx *ctx, int slot, mbk_stats *stats);
/* The host-side test behind MBK_LAZY_UNIFORM's short cut, without a device or a context: *outside = 1 when every sample
 * of the window has |c|^2 >= 4 (1 + m), m = 1e-8 (1e-4 with MBK_PRECISION_F32 in flags) -- then calc_mb_value returns 1
 * for every pixel when mrd >= 2 (WorkerCUDA.py:54-63: z1 = c^2 + c, |z1| >= |c| (|c| - 1) > 2).  For the CPU tests. */
int mbk_view_outside_circle(const mbk_view *view, uint32_t flags, int *outside);
/* Likewise: *literal = 1 when the window holds a row whose imaginary coordinate is non-zero but below 2^-900 (2^-100 in
 * binary32) -- the one case in which fma(2, zr * zi, ci) differs from the reference's fl(fl((2 * zr) * zi) + ci) (a
 * subnormal product), so that the launch takes the kernels with the literal form.  For the CPU tests. */
int mbk_view_needs_literal_doubling(const mbk_view *view, uint32_t flags, int *literal);
/* The same for a generic view / window (the multi-GPU shard unit is a row band of a view): enqueue on
 * `slot`, results land in h_counts / h_bytes (either may be NULL according to flags) after mbk_wait. */
int mbk_view_submit(mbk_ctx *ctx, int slot, const mbk_view *view, uint32_t mrd, uint32_t flags,
                    int32_t *h_counts, uint8_t *h_bytes);

/* Codec codes of DataChunkSerializer.cs (Raw :20, RLE :54). */
#define MBK_CODEC_RAW 0x00u
#define MBK_CODEC_RLE 0x01u

/* Serialise, ON THE DEVICE, the quantised bytes of the last tile this ctx computed with
 * MBK_WANT_BYTES (mbk_datachunk / mbk_view_compute): exactly the byte stream DataChunk.Serialize
 * (DataChunk.cs:173-206) writes to a chunk file / the DataServer sends to the Viewer
 * (DataServer.cs:204-220): one code byte, then the Raw payload (the n bytes) or the RLE payload
 * (repeated u32 runLength LE + u8 value, DataChunkSerializer.cs:56-100), whichever is strictly
 * shorter, Raw winning ties (serializer order, DataChunk.cs:165-168,190).  Only the serialised bytes
 * cross PCIe.  h_out must hold *size <= 1 + n bytes; cap is its capacity (MBK_ERR_INVALID if too
 * small, with *size set to the needed size). */
int mbk_serialize_last(mbk_ctx *ctx, uint8_t *h_out, uint64_t cap, uint64_t *size, uint32_t *codec);

/*
 * Tuning options.  They change scheduling only -- every value the setter accepts gives bit-identical
 * results (tests/test_gpu_parity.py::test_option_matrix_is_bit_exact) -- and the library reads NO
 * environment variable.  mbk_set_option returns MBK_ERR_INVALID for an unknown option or a value out of
 * range.  Defaults in brackets.
 */
enum mbk_option {
    MBK_OPT_ORDER = 0,     /* asm/group: workgroup order ([3] since round 4). 0 image order, 1 multiplicative permutation, 2 heavy-first list
                              (probe never escaped first; optionally a middle class, MBK_OPT_PROBE_MID), 3 "units" (round 4,
                              csrc/mbk_units.h): probe never escaped / escaped at step 4..31 one workgroup per block in that
                              order, then the blocks whose probe escaped within 3 steps eight block columns to a workgroup
                              through the light path; launches it has no form for (kernel "asm", smooth output, 64-bit
                              quantiser, block columns not a multiple of 8 or > 2048, group_steps != 16) take order 2 */
    MBK_OPT_WAVES_PER_WG,  /* asm/group: 8x8 blocks per workgroup: [1], 2, 4 */
    MBK_OPT_GROUP_STEPS,   /* group: steps per grouped bailout test: 4, 8, [16], 32 (16 / 32 apply to the blocks classified as
                              interior -- probe-heavy / dense --, the rest keep 8).  Scan pass 2 and the fp32 loops have no
                              4-step form: there 4 means 8 (and the cycle test needs >= 8, so 4 in "group" runs without it).
                              32 exists for the fp64 "group" kernel without the cycle test only; everywhere else it means 16 */
    MBK_OPT_EXACT_STEPS,   /* group / scan pass 2: steps tested one by one before the grouped test takes over: 0..4096 [8] */
    MBK_OPT_PROBE_STEPS,   /* asm/group: depth of the heavy-first probe: 2..65536 [32] */
    MBK_OPT_SCAN_WAVES,    /* scan: resident waves per SIMD of pass 1: 1..[8] */
    MBK_OPT_SCAN_XCD_MAP,  /* scan: XCD-aware block-column order (a 128-byte output line is completed in one L2): 0, [1] */
    MBK_OPT_SCAN_COL_PERIOD, /* scan: sweeps a pass-1 wave stays in one block column before jumping to a far one: 0 (never) .. 65536 [4] */
    MBK_OPT_HEAVY_SHARE,   /* default kernel: share (x 65536) of the window's probe pixels still inside after 4 steps above
                              which the launch uses "group" instead of "scan": 0 (always group) .. 65536 (always scan) [655 = 1 %] */
    MBK_OPT_RF_LIVEMIN,    /* refill: refill when this many lanes or fewer are live: 0..63 [48] */
    MBK_OPT_RF_PATIENCE,   /* refill: steps between forced refill checks: 16..2^20 [256] */
    MBK_OPT_RF_BATCH,      /* refill: blocks per queue pop: 1..64 [1] */
    MBK_OPT_RF_WAVES,      /* refill: resident waves per SIMD: 1..[8] */
    MBK_OPT_CYCLE_DETECT,  /* group / scan pass 2 (8- and 16-step groups): retire a pixel as "never escapes" as soon as its
                              (zr, zi) bit pattern repeats an earlier state of its own orbit -- the step map is a
                              deterministic function of those bits, so the reference's loop provably runs to mrd-1 and
                              returns 0.  Same counts, fewer executed steps on tiles that hold part of the set: 0, [1] */
    MBK_OPT_PROBE_MID,     /* asm/group: a block whose probe pixel escapes at step >= this value goes to a middle dispatch
                              class (after the blocks whose probe never escaped, before the rest): 2..[65537]; a value
                              above probe_steps leaves the middle class empty = the two-class order.  Measured on cfg2
                              (profiles/r03): 6 -> 581.8 us per launch, off -> 577.8: the light blocks are dispatch-bound
                              and must stay interleaved with the boundary blocks, so the default is off */
    MBK_OPT_PREPASS_OVERLAP, /* asm/group: run the dispatch-order pre-pass (one classify kernel, 13 us on cfg2) on an auxiliary
                              stream, into one of three dispatch lists used in turn, so that it overlaps the PREVIOUS launches' tile
                              kernels on the caller's stream (the tile kernel waits for its list through an event): 0 = on the
                              caller's stream, in order (no second queue, no event; +4..5 us per cfg2 launch back to back), [1],
                              2 = as 1 with the auxiliary stream at the highest priority the device offers */
    MBK_OPT_EXACT_LONG,    /* group / scan pass 2: cap on exact_steps for the blocks that run 16-step groups (classified as
                              interior, where hardly any lane escapes early -- and one that does costs a trip plus the
                              block's single fix-up): [0] = no per-step prologue for them .. 4096 (cfg2 +0.4 %, inset +0.5 %) */
    MBK_OPT_SCAN_INLINE,   /* scan: when the host's probe of the window finds no pixel that outlives the light pass (an
                              all-exterior tile: 3 in 4 of a pyramid level), pass 1 finishes whatever blocks it cannot
                              finish in 4 steps itself, on the spot, stays in its block column, and pass 2 is not launched
                              (it cost 4.4 us to find empty lists): 0, [1] */
    MBK_OPT_WAVE_LIMIT,    /* asm/group and scan pass 2: cap on the resident waves per SIMD of the one-wave-per-block kernels,
                              imposed through unused dynamic LDS: [0] = none (8), 1..7.  Round 4 measured that the SIMD
                              arbiter serves its OLDEST wave first and that 2-3 waves saturate the fp64 pipe
                              (profiles/r04/valu_issue.txt): fewer resident waves shorten the drain at the end of a
                              launch but leave fewer slots to hide the latency of light blocks */
    MBK_OPT_UNITS_MIN_LIGHT, /* order 3: the units kernel serves a launch only when at least this share (x 65536) of the host's
                              probe pixels of the window is gone after 4 steps -- where there is little light area to batch, the
                              plain one-block-per-workgroup kernel (order 2) is the leaner one (cfg3: -0.7 %): 0 (always) ..
                              65536 [32768 = one half] */
    MBK_OPT_XCD_BALANCE,   /* order 3: shares of the eight XCDs in the units kernel's list of heavy blocks.  The XCDs of one chip run
                              2-10 % apart and the hardware deals them equal numbers of workgroups, so a launch lasts as long as
                              its slowest XCD: [0] even shares (default since round 5), 1 shares that follow the time stamps
                              earlier launches on the same stream left in pinned memory (72 stores per launch; the first launch
                              on a stream is even; only launches without the cycle test, and only those that had the chip to
                              themselves, are followed: strict cfg2 +0.7..1.2 %, nothing for the library's default path with
                              the cycle test or several tiles in flight -- which is why it is opt-in: bench.py times it beside its
                              headline, as `xcd_balance_opt_in`), 2 a fixed uneven deal (tests).  Changes when a block is computed,
                              never what is stored */
    MBK_OPT_M_LATE,        /* order 3: boundary blocks (centre pixel gone within the probe's 32 steps) whose centre escapes at
                              step >= this value open the dispatch order, before the interior blocks: the ~200 of them that
                              hold a never-escaping pixel run as long as an interior block and used to start a few microseconds
                              before the dispatchers ran dry (csrc/mbk_units.h): 0 (off: H, M, V), 1..65536 [8] -- a value above
                              MBK_OPT_PROBE_STEPS leaves the class empty (the probe reports no later step); the heavy-first
                              list of the launches the units kernel does not serve uses it when it lies in 2..probe_steps.
                              Changes when a block is computed, never what is stored */
    MBK_OPT_H_SETTLED,     /* order 3, with the cycle test only: interior blocks whose probe orbit is within 10^-k of settled (min over p of
                              |z_32 - z_(32-p)|^2 of the centre pixel) retire within a few checks, the others run (nearly) all
                              steps; the settled ones are dispatched behind the others, so that the last interior blocks to
                              start are short ones: 0 (one list), k = 1..30 [6].  Measured only together with M late
                              (profiles/r05).  Changes when a block is computed, never what is stored */
    MBK_OPT_CLASSIFY_WG,   /* order 3: threads per workgroup of the probe pre-pass, a multiple of 64: 64 .. [1024].  The pre-pass of launch
                              L + 1 runs beside the tile kernel of launch L (MBK_OPT_PREPASS_OVERLAP); a 1024-thread workgroup needs
                              16 free wave slots on ONE CU at once, which a chip full of single-wave workgroups offers only in its
                              drain */
    MBK_OPT_SCAN_STRIP,    /* scan, finish-in-place form (MBK_OPT_SCAN_INLINE; windows at least 512 pixels wide): a wave's region is 64 x 1
                              pixels instead of an 8x8 block, so that every store instruction writes one contiguous 256-byte
                              (int32) / 64-byte (uint8) piece of a row -- the all-exterior tile is bound by its stores, and a
                              store-only fill of the same box writes rows 1.5x (int32) / 2.5x (uint8) faster than 8x8 blocks
                              (profiles/r05/fill.txt): 0, [1].  Same loop, same arithmetic: which lane holds which pixel */
    MBK_OPT_CYCLE_WINDOW,  /* cycle test (MBK_OPT_CYCLE_DETECT): how the window of the saved reference state grows.  A pixel whose
                              orbit becomes bitwise periodic at step s retires at the first reference state taken after s (plus
                              lcm(8, period) steps); doubled windows (0: rounds 2-4) take them at steps 8, 16, 32, 64, ... -- 1.44 s
                              on average.  [32]: the window grows by a quarter (+ 1) while it is shorter than this many 8-step
                              checks and doubles from there on (long periods -- deep zooms -- need long windows): 6-7 % fewer
                              wave-steps on full-set views and shallow DataChunks, cfg3 unchanged (scripts/cycle_window_model.c).
                              0 .. 65536.  Every schedule is exact: a bitwise repeat proves the cycle whichever two steps match */
    MBK_OPT_SPILL_FIRST,   /* group, launches the units kernel does not serve (deep zooms; fp64 with 16-step groups, fp32; counts / bytes;
                              single-wave workgroups; dispatch order 2 or 3): SPILL (round 6, csrc/mbk_kernels.h block_pixel_spill,
                              csrc/mbk_spill.h).  A wave runs until its last lane is done, and on a deep zoom a third of the blocks
                              reach step ~500 with a handful of their 64 lanes alive.  At checkpoints -- this many steps after the
                              per-step prologue, then twice as far each time -- a block with at most MBK_OPT_SPILL_LANES live lanes writes
                              their state to a list in HBM and ends; a prefix sum compacts the list (no atomics) and a second kernel
                              runs the listed lanes 64 to a wave from where they stopped.  Same recurrence from the same state:
                              identical counts.  0 = off, else a multiple of 32 up to 65536 [256] */
    MBK_OPT_SPILL_LANES,   /* SPILL: a block spills when this many lanes or fewer are alive at a checkpoint (= the slots of 24 bytes a
                              block owns: state, lane / step word, list entry -- 403 MB for an 8192^2 window at 16): 1 .. 32 [16] */
    MBK_OPT_SPILL_MIN_MRD, /* SPILL: only launches with mrd at least this deep (shallow tiles have nothing to hand over): [2048] */
    MBK_OPT_SPILL_MIN_BLOCKS, /* SPILL: ... and only launches of at least 2^this 8x8 blocks: [19] = 5 800^2 pixels.  The second pass cannot
                              be shorter than one wave running the steps a never-escaping pixel has left (~ mrd x 21 ns: 0.2 ms at
                              mrd 10 000, 1 ms at 50 000) plus four small kernels (~30 us), while the first pass' saving grows with
                              the launch: below ~2^19 blocks the pass costs what it saves.  0 .. 31 */
    MBK_OPT_SPILL_CYC_SHIFT, /* SPILL with the cycle test: the second pass resumes orbits that have n steps behind them; the window of its
                              first reference state is n >> this checks (of 8 steps) instead of 1 -- [5]: where the schedule of an
                              unbroken run would stand (windows of ~ n / 4 steps); 31 = start at 1.  Any schedule is exact.  0 .. 31 */
    MBK_OPT_COUNT_
};
/* Read-only diagnostics through mbk_get_option: what hipOccupancyMaxActiveBlocksPerMultiprocessor reports for the
 * four scan-path kernels, in single-wave workgroups per CU: +0 f64 scan, +1 f64 heavy, +2 f32 scan, +3 f32 heavy. */
#define MBK_INFO_SCAN_WG_PER_CU 100
/* MBK_OPT_XCD_BALANCE = 1, of the stream that has reported most: +0..+7 the share of XCD x of the heavy list (x 2^20; an even
 * deal is 131072), +8 the number of the last launch whose time stamps were read, +9 the units launches issued. */
#define MBK_INFO_XCD_SHARE 110
/* SPILL (MBK_OPT_SPILL_FIRST): +0 the lanes the last launch with a second pass handed over to it (waits for that launch),
 * +1 the number of launches of this ctx that ran with one. */
#define MBK_INFO_SPILL 130
int mbk_set_option(mbk_ctx *ctx, int option, uint32_t value);
int mbk_get_option(mbk_ctx *ctx, int option, uint32_t *value);

/* Diagnostics of the units kernel's deal across the eight XCDs (MBK_OPT_XCD_BALANCE), on the host, without a device or a
 * context: the functions the kernels call.  mbk_units_plan: the shares for a list of n_h heavy entries, n_m middle entries
 * and n_v row units under the H fractions `fractions[8]` (sum 1) -> plan[40] ([2] = number of workgroup ids, [8 + x] /
 * [16 + x] = heavy / light entries of XCD x).  mbk_units_lookup: what workgroup id `id` computes: *list = 0 nothing, 1 heavy
 * entry *index, 2 middle entry *index, 3 row unit *index.  Every entry of every list is taken by exactly one id. */
int mbk_units_plan(uint32_t n_h, uint32_t n_v, uint32_t n_m, const double *fractions, uint32_t *plan);
int mbk_units_lookup(const uint32_t *plan, uint32_t id, uint32_t *list, uint32_t *index);

/* The quantiser alone, on the device: h_bytes[i] = uint8(ceil(h_counts[i] * 256 / mrd)) for n host counts
 * (WorkerCUDA.py:96-98; each count must lie in [0, mrd-1], which is what calc_mb_value returns).
 * Exists so that the exactness of the device's division-free form can be checked for every count. */
int mbk_quantise_counts(mbk_ctx *ctx, const int32_t *h_counts, uint64_t n, uint32_t mrd, uint8_t *h_bytes);

/* Device-side reduction over int32 counts already in HBM (asynchronous part on hip_stream -- NULL =
 * the null stream -- then a stream sync): fills stats->pixel_iterations and stats->never_pixels.  Used by bench.py to turn
 * kernel time into pixel-iterations/s from the kernel's own output. */
int mbk_reduce_counts(mbk_ctx *ctx, const int32_t *d_counts, uint64_t n, uint32_t mrd,
                      void *hip_stream, mbk_stats *stats);

/*
 * The worker loop in native code: lease -> compute -> send, pipelined, until the Distributer answers 0x11
 * ("no workload available") or max_tiles (0 = no limit) have been leased.  Replaces the loop of the reference worker,
 * WorkerCUDA.py:111-184 (do_workload_single called from main until it returns False), speaking the protocol of
 * Distributer.cs:30-45,358-458 / DistributerWorkload.cs:53-100 UNCHANGED: per tile the wire sees exactly the
 * reference's two exchanges (request 0x00 -> 0x10 + 4 x u32 | 0x11; response 0x01 + 4 x u32 -> 0x20 | 0x21, then on
 * 0x20 exactly MBK_CHUNK_BYTES raw bytes); only their timing overlaps with other tiles': while tile n is on the GPU
 * (slot n % MBK_SLOTS, its D2H overlapping the other slots' kernels) tile n+1 is being leased and tiles <= n-1 are being sent
 * by `senders` threads (1..64) on their own connections.  `senders + MBK_SLOTS` pinned 16 MiB buffers circulate, so a slow
 * server back-pressures the lease rate.  Uniform tiles (all 0 / all 1) are not copied off the GPU: their payload
 * comes from a shared constant buffer.  A rejected tile (0x21) is dropped and the loop carries on (WorkerCUDA.py:
 * 161-163).  On a socket error while leasing the loop stops leasing, finishes the tiles it holds, and returns
 * MBK_ERR_NET (mbk_last_error has the text).  The same single-host-thread rule as everything else on a ctx: the
 * calling thread drives the GPU; the sender threads touch sockets and host buffers only.
 * distributedmandelbrot_amd/worker.py: run_native / run_farm bind it; run_pipelined is the same loop in Python.
 */
typedef struct mbk_worker_report {
    uint64_t leased;           /* tiles obtained with opcode 0x00 */
    uint64_t accepted;         /* 0x20 and every payload byte handed to the socket */
    uint64_t rejected;         /* 0x21 */
    uint64_t resets;           /* 0x20, then the server reset the connection mid-payload (the reference server reads the
                                  payload with ONE Receive and closes, Distributer.cs:416-423: the tile is complete there) */
    uint64_t uniform_tiles;    /* sent from the shared constant buffer */
    uint64_t pixel_iterations; /* reference-equivalent, summed over the tiles */
    double kernel_ms_sum;
    double seconds;            /* wall time of the call */
    uint64_t net_retries;      /* exchanges that were repeated after a transient network failure (round 4, ABI 4) */
} mbk_worker_report;
int mbk_worker_run(mbk_ctx *ctx, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
                   mbk_worker_report *report);

/*
 * Process-wide network behaviour of mbk_worker_run / mbk_feeder_run (every feeder of a farm shares it; round 4).  Why:
 * the reference Distributer accepts on ONE thread, one connection at a time, with a listen backlog of 16
 * (Distributer.cs:16,221,226-297) and 100 ms receive timeouts (:17,196-202), and the reference worker opened one
 * connection at a time (WorkerCUDA.py:115,148).  8 feeders x (4 senders + 1 lease connection) must not present that
 * server with 40 concurrent connects -- a full backlog is an RST on a Windows/.NET host -- and a computed tile must
 * not be dropped because one connect failed.  Both exchanges are retried with exponential backoff on transient
 * failures (refused, reset, timed out, closed before the reply) until the server has answered; after 0x20 the payload
 * is sent once (the server removed the lease when it accepted, Distributer.cs:404-423).
 */
enum mbk_net_option {
    MBK_NET_MAX_CONNECTIONS = 0,    /* connections open or being opened at any time, whole process; 1..64, default 8 */
    MBK_NET_CONNECT_TIMEOUT_MS = 1, /* 0 = wait for ever; default 10 000 */
    MBK_NET_IO_TIMEOUT_MS = 2,      /* per send / receive call without progress; 0 = none; default 30 000 */
    MBK_NET_RETRIES = 3,            /* attempts after the first, per exchange; 0..100, default 6 */
    MBK_NET_BACKOFF_MS = 4,         /* first pause; doubles per attempt up to 2 s, plus jitter; 1..10 000, default 50 */
    MBK_NET_STOP = 5,               /* 1: every running loop stops leasing, returns the tiles it holds and ends (Ctrl-C
                                       handlers set it from another thread); 0 re-arms */
    MBK_NET_PEAK_CONNECTIONS = 6,   /* read-only (mbk_net_get_option): the most connections that were ever open at once;
                                       setting MBK_NET_MAX_CONNECTIONS clears it */
    MBK_NET_FEEDER_SLOTS = 7,       /* tiles mbk_feeder_run keeps on its backend at once (slot numbers 0 .. k-1); 1..8, default 2 */
    MBK_NET_OPT_COUNT_
};
int mbk_net_set_option(int option, uint32_t value);
int mbk_net_get_option(int option, uint32_t *value);

/* The same protocol loop over a caller-supplied compute backend (mbk_worker_run is this with the backend bound to a
 * GPU context: submit = mbk_datachunk_submit_ex(MBK_LAZY_UNIFORM), wait = mbk_wait, alloc/release = pinned memory).
 * For hosts that schedule the GPU themselves, and for the CPU tests of the protocol engine.  submit / wait are
 * called from the calling thread only, with slot cycling through 0 .. k-1 (k = MBK_NET_FEEDER_SLOTS, default 2; mbk_worker_run uses MBK_SLOTS); wait must fill stats (all_bytes_zero /
 * all_bytes_one decide whether h_bytes or a constant buffer is sent).  on_tile (optional) is called from a sender
 * thread once per returned tile with status 1 accepted, 0 rejected, 2 reset after 0x20, -1 error. */
typedef struct mbk_feeder_ops {
    void *user;
    int (*submit)(void *user, int slot, uint32_t level, uint32_t mrd, uint32_t index_real, uint32_t index_imag, uint8_t *h_bytes);
    int (*wait)(void *user, int slot, mbk_stats *stats);
    void *(*alloc)(void *user, uint64_t bytes);
    void (*release)(void *user, void *ptr);
    void (*on_tile)(void *user, const uint32_t workload[4], const mbk_stats *stats, int status);
} mbk_feeder_ops;
int mbk_feeder_run(const mbk_feeder_ops *ops, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
                   mbk_worker_report *report);

#ifdef __cplusplus
}
#endif
#endif /* MBK_H */
