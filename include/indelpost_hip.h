/*
 * indelpost_hip.h -- C ABI of libindelpost_hip.so, the MI355X drop-in for indelPost's striped
 * Smith-Waterman realignment path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Two layers:
 *
 *  (1) The reference's own FFI, symbol for symbol.  These are exactly the four functions
 *      indelpost/sswpy.pyx binds (`cdef extern from "ssw.h"`, sswpy.pyx:57-83):
 *          ssw_init      replaces ssw.h:86   / ssw.c:787-808
 *          init_destroy  replaces ssw.h:91   / ssw.c:810-814
 *          ssw_align     replaces ssw.h:126-134 / ssw.c:816-920
 *          align_destroy replaces ssw.h:139  / ssw.c:922-925
 *      Same signatures, same s_align layout (ssw.h:55-66), same ownership (result and cigar are
 *      malloc'd by the callee and freed by align_destroy; the profile BORROWS read and mat,
 *      ssw.c:803-804), same NULL-on-error convention (ssw.c:848-859).  Each call runs one
 *      alignment on GPU 0 -- correct but latency-bound; it exists so the reference's binding links
 *      unchanged.  The inline cigar helpers of ssw.h:171-192 (to_cigar_int, cigar_int_to_op, cigar_int_to_len) and the
 *      encoded_ops table they use (ssw.h:34) are provided below under the same names.
 *
 *  (2) The batched entry points that the reference's per-read loop
 *      (localn.pyx:47-66, 464-472 -> sswpy.pyx:149-178, 199-225) collapses into: a whole job table
 *      (read, window id, gap_open, gap_extension) per call.  All buffers are caller-owned.
 *
 * Every function fails loudly (negative return / NULL + ipx_last_error()) when no GPU is present;
 * there is no CPU fallback.
 */
#ifndef INDELPOST_HIP_H
#define INDELPOST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * (1) reference-compatible single-alignment interface
 * ------------------------------------------------------------------------------------------- */
struct _profile;
typedef struct _profile s_profile;            /* ssw.h:36-37 */

typedef struct {                              /* ssw.h:55-66 */
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t *cigar;
    int32_t cigarLen;
    uint16_t flag;
} s_align;

s_profile *ssw_init(const int8_t *read, const int32_t readLen, const int8_t *mat, const int32_t n,
                    const int8_t score_size);
void init_destroy(s_profile *p);
s_align *ssw_align(const s_profile *prof, const int8_t *ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters,
                   const int32_t filterd, const int32_t maskLen);
void align_destroy(s_align *a);

#define IPX_MAPSTR "MIDNSHP=X"
#ifndef MAPSTR
#define MAPSTR IPX_MAPSTR                                   /* ssw.h:29 */
#endif
#ifndef BAM_CIGAR_SHIFT
#define BAM_CIGAR_SHIFT 4u                                  /* ssw.h:30-32 */
#endif
/* ASCII CIGAR letter -> BAM opcode 0..8, everything else 0 (ssw.h:34, table ssw.c:127): defined by the library */
extern const uint8_t encoded_ops[];

static inline uint32_t to_cigar_int(uint32_t length, char op_letter)   /* ssw.h:171-173 */
{
    return (length << BAM_CIGAR_SHIFT) | (encoded_ops[(int)op_letter]);
}
static inline char cigar_int_to_op(uint32_t cigar_int)      /* ssw.h:182-184 */
{
    return (cigar_int & 0xfU) > 8 ? 'M' : IPX_MAPSTR[cigar_int & 0xfU];
}
static inline uint32_t cigar_int_to_len(uint32_t cigar_int) /* ssw.h:190-192 */
{
    return cigar_int >> BAM_CIGAR_SHIFT;
}

/* ---------------------------------------------------------------------------------------------
 * (2) batched interface
 * ------------------------------------------------------------------------------------------- */
typedef struct ipx_ctx ipx_ctx;               /* one per GPU: stream, HBM-resident batch, workspace */

/* One record per job: the fields of s_align with the cigar pointer replaced by (offset, length)
 * into the caller's uint32 cigar pool (BAM encoding len<<4|op, M=0 I=1 D=2, ssw.h:171-173). 32 B. */
typedef struct {
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t cigar_off;
    uint16_t cigar_len;   /* 0: no cigar (reference: cigar == NULL) */
    uint8_t flag;         /* s_align.flag: 0, 1 (traceback failed), 2 (path may miss a part) */
    uint8_t mode;         /* 0: 8-bit pass result, 1: 16-bit pass result, 2: reference returns NULL */
} ipx_result;

enum {
    IPX_OK = 0,
    IPX_ERR_NO_DEVICE = -1,     /* no usable GPU / HIP runtime error: see ipx_last_error() */
    IPX_ERR_ARG = -2,
    IPX_ERR_READ_TOO_LONG = -3, /* a read is longer than 4 096 bp (IPX_LONG_MAX_READ) */
    IPX_ERR_REF_TOO_LONG = -4,  /* a window is longer than 32 000 bp (IPX_MAX_REFLEN) */
    IPX_ERR_CIGAR_POOL = -5,    /* caller's cigar pool too small */
    IPX_ERR_INTERNAL = -6
};

int ipx_device_count(void);
ipx_ctx *ipx_create(int device);
void ipx_destroy(ipx_ctx *c);
const char *ipx_last_error(void);

/* Scoring and ssw_align control arguments shared by the whole batch.
 * mat: 5x5 row-major substitution matrix over A,C,G,T,N (sswpy.pyx:306-336);
 * flag/filters/filterd: as ssw_align (ssw.c:821-823); score_size: as ssw_init (ssw.c:793-802). */
int ipx_set_params(ipx_ctx *c, const int8_t *mat, int flag, int filters, int filterd, int score_size);

/* Speed-only routing switches (bit mask; 0 = everything on).  Results never depend on them -- each proof or kernel
 * variant they disable has an exact fallback -- which is what the test suite uses them for:
 *   1 no 16-bit pass before the 8-bit one    2 LDS-staged profile instead of register selectors
 *   4 no upper-bound (bracket) stage         8 one traceback launch per band width
 *  16 column maxima in global scratch instead of LDS
 *  32 16-bit passes (and the 8-bit bracket stages) in packed integers even where packed halves are exact
 *  64 half-precision passes column by column with lazy-F instead of as a wavefront over the SSE lanes
 * 128 8-bit lower-bound stage in the reference's 16-lane layout (8 reads per wave) instead of two lanes per GPU lane
 * 256 the r02 order of the 8-bit passes (lower bound, upper bound, stepped) instead of the plain recurrence first + proofs
 * 512 one launch per read-length class (no rare class listed under the next populated one)
 *1024 one launch per class instead of one per occupancy tier (wavefront kernels and the stepped 8-bit passes)
 *2048 what a proof leaves open takes the lower-bound stage before the stepped pass
 *4096 small batches too take the lane-per-job traceback kernels      8192 every CIGAR through banded_sw's DP (no ungapped shortcut)
 *16384 no anti-diagonal traceback tiers     32768 no latency tier for small batches     65536 (testing) latency tier forced
 *131072 latency tier keeps the lane-per-read overflow proof     524288 latency tier launches every pass (no speculation on empty ones)
 *262144 reads of 505 bp and more take the transcribed loops (k_dp_long) in the 16-bit passes too, not one wavefront per read (k_dp_wide)
 *4194304 the latency-bound kernels keep the default wave priority     8388608 big batches: one lane-per-job traceback launch per band width
 *16777216 the reverse passes launch class c - 1 beside every forward class c
 * (csrc/ipx_types.h IPX_ROUTE_* is the authoritative list) */
int ipx_set_routing(ipx_ctx *c, int flags);

/* Stage a job table in HBM.  reads/refs: concatenated int8 codes (0..4); read_off: n_jobs+1,
 * ref_off: n_refs+1 offsets; ref_id[j]: window of job j; gap_open/gap_ext: already narrowed to
 * uint8 (ssw.h:129-130); mask_len: per job, or NULL for max(15, readLen/2) (sswpy.pyx:209-211). */
int ipx_upload(ipx_ctx *c, const int8_t *reads, const int64_t *read_off, const int8_t *refs,
               const int64_t *ref_off, const int32_t *ref_id, const uint8_t *gap_open,
               const uint8_t *gap_ext, const int32_t *mask_len, int64_t n_jobs, int32_t n_refs);
int ipx_run(ipx_ctx *c);                      /* enqueue the whole pipeline on the context's stream */
int ipx_sync(ipx_ctx *c);                     /* wait; returns IPX_OK or the first error of the run */
int ipx_download(ipx_ctx *c, ipx_result *out, uint32_t *cigar_pool, int64_t cigar_cap, int64_t *n_cigar_ops);

/* Transfers that do not wait.  ipx_set_async_io(c, 1): the caller promises to keep the buffers it passes to ipx_upload
 * valid and unchanged until the next ipx_sync, so ipx_upload only enqueues its copies (truly asynchronous when the buffers
 * are page-locked: ipx_pin_host / ipx_unpin_host wrap hipHostRegister for caller-owned memory).  ipx_download_async, after
 * ipx_sync, enqueues the copies of the records and of the n_cigar_ops CIGAR ops of the last run into caller buffers that
 * must stay valid until ipx_wait returns. */
int ipx_set_async_io(ipx_ctx *c, int on);
int ipx_pin_host(void *p, int64_t bytes);
int ipx_unpin_host(void *p);
int ipx_download_async(ipx_ctx *c, ipx_result *out, uint32_t *cigar_pool, int64_t cigar_cap, int64_t *n_cigar_ops);
int ipx_wait(ipx_ctx *c);

/* upload + run + sync + download in one call */
int ipx_align_batch(ipx_ctx *c, const int8_t *reads, const int64_t *read_off, const int8_t *refs,
                    const int64_t *ref_off, const int32_t *ref_id, const uint8_t *gap_open,
                    const uint8_t *gap_ext, const int32_t *mask_len, int64_t n_jobs, int32_t n_refs,
                    ipx_result *out, uint32_t *cigar_pool, int64_t cigar_cap, int64_t *n_cigar_ops);

/* measurement: HIP events on the context's stream */
int ipx_set_profiling(ipx_ctx *c, int on);   /* 0 off, 1 every kernel launch, 2 only the striped DP kernels */
int ipx_num_kernel_classes(void);
const char *ipx_kernel_class_name(int k);
int ipx_kernel_times(ipx_ctx *c, float *ms, int *launches);   /* arrays of ipx_num_kernel_classes() */
/* alignments processed by the launches of each striped-DP kernel class since profiling was switched on (planner tile
 * counts x alignments per tile; 0 for the other classes): the "units one launch processes" of the roofline */
int ipx_kernel_units(ipx_ctx *c, int64_t *units);
float ipx_last_run_ms(ipx_ctx *c);            /* events around the last ipx_run, valid after ipx_sync */
int ipx_debug_reruns(ipx_ctx *c);             /* runs ipx_sync had to repeat because a pass predicted empty (small batches: not launched) held a job (diagnostic) */
int ipx_debug_tb_counts(ipx_ctx *c, uint32_t *out12);  /* traceback routing of the last run (diagnostic): jobs per list -- [0..6] lane-per-job kernels (band 1..7), [7] wave-per-job kernel, [8..10] anti-diagonal tiers of 16 / 32 / 64 lanes per job, [11] 0 */

/* host-side helper: the CIGAR strings of a whole batch in one call, formatted as sswpy.pyx:283-289 does per
 * alignment ("%d%c" per op, op letters MIDNSHP=X, anything above 8 -> 'M').  Strings are concatenated into `out`
 * (no terminators); off[i]..off[i+1] delimits job i (empty for a job without CIGAR), off has n+1 entries.
 * Returns the total length, or -(needed length) when `cap` is too small (nothing is written then). */
int64_t ipx_format_cigars(const ipx_result *rec, const uint32_t *cigar_pool, int64_t n, char *out, int64_t cap, int64_t *off);
/* host-side helper, many loci as ONE batch: concatenate n_tables per-locus job tables (the arrays ipx_upload takes, one set per locus).
 * desc: n_tables x 10 int64 = addresses of reads, read_off, refs, ref_off, ref_id, gap_open, gap_ext, mask_len (0: none), then n_jobs, n_refs.
 * ipx_concat_sizes: totals[4] = read bytes, window bytes, jobs, windows; returns 1 when every table has a mask_len.  ipx_concat_tables fills
 * caller-allocated arrays (read_off / ref_off with one entry more than jobs / windows): jobs in table order, windows renumbered; IPX_OK or
 * IPX_ERR_ARG.  Replaces nothing in the reference: it is what makes one call per MANY loci cheap where the reference makes one per read
 * (localn.pyx:47-66). */
int ipx_concat_sizes(const int64_t *desc, int64_t n_tables, int64_t *totals);
int ipx_concat_tables(const int64_t *desc, int64_t n_tables, int8_t *reads, int64_t *read_off, int8_t *refs, int64_t *ref_off, int32_t *ref_id,
                      uint8_t *gap_open, uint8_t *gap_ext, int32_t *mask_len);

/* Host-side helper: the jobs of a table grouped by read length (stable).  order[k] = the job that comes k-th (n_jobs entries); the read letters,
 * read_off_out (n_jobs + 1) and the per-job arrays in that order; mask_len / mask_len_out may be NULL.  The windows are not touched.  A batch of mixed
 * lengths that stays resident is cut into stream slices after this (a slice then holds one or two length classes: config 4 58 -> 65 M alignments/s);
 * results come back in the grouped order, order[] takes them home. */
int ipx_group_by_length(const int8_t *reads, const int64_t *read_off, const int32_t *ref_id, const uint8_t *gap_open, const uint8_t *gap_ext,
                        const int32_t *mask_len, int64_t n_jobs, uint32_t *order, int8_t *reads_out, int64_t *read_off_out, int32_t *ref_id_out,
                        uint8_t *gap_open_out, uint8_t *gap_ext_out, int32_t *mask_len_out);
/* host-side helper: FNV-1a (32 bit) of every job's BAM-encoded CIGAR ops, 2166136261 for a job without CIGAR (whole batches are
 * compared op for op with the reference through these, and record digests are built on them) */
void ipx_cigar_hashes(const ipx_result *rec, const uint32_t *cigar_pool, int64_t n, uint32_t *out);
/* host-side helper: 64-bit digest (XXH64, seed 0) of a batch's results in job order -- per job ten little-endian int64: score1, score2,
 * ref_begin1, ref_end1, read_begin1, read_end1, ref_end2, flag, cigar_len, cigar_hash[i] (from ipx_cigar_hashes).  bench.py compares it
 * with the digest of the reference's results on the same job table (tests/golden/bench_digests.json). */
uint64_t ipx_record_digest(const ipx_result *rec, const uint32_t *cigar_hash, int64_t n);

/* deterministic synthetic workload of SURVEY.md section 8d (xorshift64), host side:
 * one window of `wl` codes and n reads of `rl` codes; returns the final generator state */
uint64_t ipx_synth_window(uint64_t state, int8_t *ref, int32_t wl);
uint64_t ipx_synth_reads(uint64_t state, const int8_t *ref, int32_t wl, int8_t *reads, int64_t n, int32_t rl);
/* the same generator over many windows: n_windows windows with lengths drawn from
 * [wl_lo, wl_hi]; per window `per` reads of each length in rls[0..n_rls) (clamped to the window length).  Caller
 * buffers sized for the worst case; returns the number of jobs, -1 on a bad argument; *state is advanced. */
int64_t ipx_synth_mixed(uint64_t *state, int32_t n_windows, int32_t wl_lo, int32_t wl_hi, const int32_t *rls,
                        int32_t n_rls, int32_t per, int8_t *refs, int64_t *ref_off, int8_t *reads,
                        int64_t *read_off, int32_t *ref_id);

#ifdef __cplusplus
}
#endif
#endif
