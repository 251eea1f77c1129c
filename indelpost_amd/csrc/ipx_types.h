// ipx_types.h -- device-side data model of one alignment batch (shared by the HIP runtime, the
// kernels and the test-only emulator).  Vocabulary follows the reference: read, reference window,
// gap open/extension, mask length, score1/score2, CIGAR (ssw.h:55-66, 126-134).
#pragma once
#include <stdint.h>

// One result record per job: the fields of s_align (ssw.h:55-66) with the malloc'd cigar pointer
// replaced by (offset,length) into a caller-owned uint32 pool.  32 bytes.
struct IpxResult {
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t cigar_off;  // first op in the cigar pool (valid when cigar_len > 0)
    uint16_t cigar_len;  // 0 = no cigar (reference: cigar == NULL)
    uint8_t flag;        // 0 ok, 1 traceback failed, 2 reverse score < score1 (ssw.c:888-891, 911)
    uint8_t mode;        // internal: see IPX_MODE_*
};

enum {
    IPX_MODE_BYTE = 0,       // score1 came from the 8-bit pass
    IPX_MODE_WORD = 1,       // 8-bit pass overflowed (score 255), 16-bit pass used (ssw.c:844-847)
    IPX_MODE_FAIL = 2,       // reference would return NULL (8-bit only profile overflowed, ssw.c:848-851)
    IPX_MODE_NEED_WORD = 3,  // 8-bit pass overflowed, 16-bit pass still to run
    IPX_MODE_NEED_BYTE_EXACT = 4,  // 8-bit bracket inconclusive, exact (stepped) 8-bit pass still to run
    IPX_MODE_WORD_UNPROVEN = 5,    // 16-bit result computed FIRST; 8-bit overflow not yet established
    IPX_MODE_NEED_BYTE_CHECK = 6,  // ... and not provable from the diagonal: the 8-bit pass decides
    IPX_MODE_NEED_BYTE_EXACT_W = 7,// as NEED_BYTE_EXACT, with a 16-bit result already in the record
    IPX_MODE_NEED_BYTE_HIGH = 8,   // the record holds the LOWER-bound stage's outputs; the upper-bound stage certifies them or not
    // "plain first" flow of the 8-bit passes (r03, see IPX_PASS_BYTE_FIRST): the plain recurrence runs FIRST and is certified afterwards
    IPX_MODE_BYTE_PLAIN = 9,       // forward 8-bit result final AND equal to the plain recurrence's: the reverse pass may run as a wavefront
    IPX_MODE_NEED_FWD_PROOF = 10,  // the record holds the plain recurrence's outputs; the best cell still needs its lower-bound proof
    IPX_MODE_NEED_FWD_PROOF2 = 11, // ... and so does the second-best column
    IPX_MODE_NEED_BYTE_LOW = 12,   // plain recurrence reached the overflow threshold: the lower-bound stage decides (nothing to compare with)
    IPX_MODE_NEED_BYTE_LOW_CMP = 13,  // proof failed: the lower-bound stage runs and is compared with the plain outputs in the record
    IPX_MODE_NEED_REV_PROOF = 14,
    IPX_MODE_NEED_BYTE_EXACT_P = 15,  // as NEED_BYTE_EXACT with the plain recurrence's outputs in the record: equal exact outputs keep the plain reverse pass  // begin position from the plain reverse recurrence in the record; its cell still needs the proof
    IPX_MODE_BYTE_OPT = 16,  // final 8-bit record (both passes done) whose score1 is KNOWN to be the plain recurrence's optimum (certified by k_prove_plain);
                             //   transient: k_tb_list, its only reader (the ungapped shortcut needs an exact optimum), turns it into IPX_MODE_BYTE
    IPX_MODE_PENDING = 255,  // not processed yet
};

// Passes of the pipeline.  A pass is a job list bucketed by class (= striped segment count, + IPX_SLOW_BASE for jobs with
// gap_open <= gap_ext, which need kernels with the stepped lazy-F loop).  STATIC passes depend only on host-known facts
// (read lengths, penalties, scoring parameters): their lists are built once per resident batch.  The lists of the other
// passes are decided on the device: the kernel that decides a job's next pass also counts it (k_dp_pass finalisation,
// k_prove_overflow), so that one scatter launch per pass is all the planning left on the stream.
enum {
    IPX_PASS_WORD_FIRST = 0,      // static: 16-bit forward pass BEFORE the 8-bit one, for reads that will almost surely overflow
    IPX_PASS_BYTE_FIRST = 1,      // static: the first 8-bit stage of every other read.  Bracket flow (r02): the lower-bound stage.  Plain-first
                                  //   flow (r03, IpxBatch::plain_first): the plain recurrence as a wavefront (k_dp_skew BH = 2), an upper bound of
                                  //   the 8-bit matrix cell by cell, certified afterwards by k_prove_plain (a banded lower bound through the best cell)
    IPX_PASS_BYTE_CHECK = 2,      // 8-bit lower-bound stage for word-first reads whose overflow could not be proven
    IPX_PASS_BYTE_HIGH = 3,       // 8-bit forward pass, upper-bound stage: certifies the lower-bound outputs
    IPX_PASS_BYTE_EXACT = 4,      // 8-bit forward pass with the reference's stepped lazy-F: what the bracket left open, and (a static
                                  //   part, counted once per batch) short reads that can neither overflow nor profit from the bracket
    IPX_PASS_WORD_FWD = 5,        // 16-bit forward pass after an 8-bit overflow (ssw.c:844-847)
    IPX_PASS_BYTE_REV = 6,
    IPX_PASS_WORD_REV = 7,
    IPX_PASS_BYTE_LOW2 = 8,       // plain-first flow: lower-bound stage for the reads whose plain result could not be certified by proof
    IPX_PASS_BYTE_REV_PLAIN = 9,  // plain-first flow: 8-bit reverse pass as the plain recurrence (reads in IPX_MODE_BYTE_PLAIN), certified by proof
    IPX_NUM_PASSES = 10,
    IPX_FIRST_DYNAMIC_PASS = 2,
    IPX_PASS_MC_LDS = 0x100,      // flag or-ed into a DP kernel's `pass` argument: column maxima live in LDS (room reserved by the launch)
};

// stage of an 8-bit forward kernel (template parameter of k_dp_pass)
enum { IPX_STAGE_EXACT = 0, IPX_STAGE_LOW = 1, IPX_STAGE_HIGH = 2 };

// speed-only routing switches (ipx_set_routing): every combination must give identical results, the tests run them all
enum {
    IPX_ROUTE_NO_WORD_FIRST = 1,   // never run the 16-bit pass before the 8-bit one
    IPX_ROUTE_NO_PERM_PROFILE = 2, // LDS-staged int8 profile instead of the register selectors
    IPX_ROUTE_NO_BRACKET = 4,      // no upper-bound stage: a read the lower-bound stage cannot settle goes to the stepped pass
    IPX_ROUTE_TB_NO_FUSE = 8,      // one traceback launch per band width even for small batches
    IPX_ROUTE_NO_MC_LDS = 16,      // column maxima in the global scratch even when they would fit in LDS
    IPX_ROUTE_NO_F16 = 32,         // 16-bit passes in packed integer arithmetic even where the half-precision form applies
    IPX_ROUTE_NO_SKEW = 64,        // half-precision 16-bit passes column by column with lazy-F (k_dp_pass) instead of as a wavefront (k_dp_skew)
    IPX_ROUTE_NO_VL2 = 128,        // 8-bit lower-bound stage in the reference's 16-lane layout (8 reads per wave) instead of two reference
                                   //   lanes per GPU lane (16 reads per wave)
    IPX_ROUTE_NO_PLAIN_FIRST = 256,  // 8-bit passes in the r02 bracket order (lower bound, upper bound, stepped) instead of plain recurrence + proof
    IPX_ROUTE_NO_CLASS_MERGE = 512,  // every segLen class keeps its own wavefront launch (no rare class served by a longer class's kernel)
    IPX_ROUTE_NO_TIERS = 1024,       // one launch per class of the stepped 8-bit passes too (no k_dp_pass_tier).  (r03: also the wavefront passes' tier launches, gone in r04)
    IPX_ROUTE_NO_EXACT_DIRECT = 2048, // a read the proofs leave open takes the lower-bound stage before the stepped one (r03 first half) instead of
                                      //   the stepped pass at once (which steps only where a cut can happen and costs little more than the lower bound)
    IPX_ROUTE_TB_NO_WAVE_PER_JOB = 4096,  // small batches too take the lane-per-job traceback kernels (default: up to 2048 jobs, one wave per job)
    IPX_ROUTE_TB_NO_UNGAPPED = 8192,      // every CIGAR through banded_sw's DP (default: an alignment whose diagonal alone reaches score1 gets its one-run CIGAR from k_tb_list)
    IPX_ROUTE_TB_NO_DIAG = 16384,         // no anti-diagonal traceback tiers (k_tb_diag): bands wider than 7, doubled bands and small batches take one wave per job (k_tb_coop) as in r03
    IPX_ROUTE_NO_LAT = 32768,             // small batches too take the 8-lanes-per-read wavefront kernels (no latency tier: k_dp_skew W = 32)
    IPX_ROUTE_NO_LAT_PROOF = 131072,      // the latency tier keeps the lane-per-read overflow proof (k_prove_overflow) instead of k_prove_overflow_diag
    IPX_ROUTE_NO_SPECULATE = 524288,      // the latency tier launches every pass, also those the previous run found empty
    IPX_ROUTE_TEST_SKIP_ALL = 1048576 * 2, // (testing) every dynamic pass of the latency tier is predicted empty: the guard must notice and the run be repeated
    IPX_ROUTE_NO_WIDE = 262144,           // reads of 64 segments or more take the transcribed loops (k_dp_long) in the 16-bit passes too (default: one wavefront per read, k_dp_wide)
    IPX_ROUTE_NO_SETPRIO = 4194304,       // the latency-bound kernels keep the default wave priority
    IPX_ROUTE_TB_PER_WIDTH = 8388608,     // big batches: one lane-per-job traceback launch per band width 1, 2, 3 and one for 4..7, a doubled band served by the next width's launch (r02..r04; default: one launch, its blocks shared out on the device)
    IPX_ROUTE_REV_BELOW = 16777216,       // the reverse passes launch class c - 1 beside every forward class c (r03; default: prefixes of class c - 1 ride in c's launch)
    IPX_ROUTE_NO_BAND_REV = 33554432,     // the 16-bit reverse pass over the whole prefix rectangle for every job (default: as a band where the job's score budget allows)
    IPX_ROUTE_FORCE_BAND_REV = 67108864,  // (testing) the banded reverse pass whatever the size of the class
    IPX_ROUTE_FORCE_LAT = 65536,          // (testing) the latency tier whatever the batch size, where its other conditions hold
    IPX_ROUTE_INTERNAL_VL2 = 1 << 20,   // (set by ipx_run_pipeline itself: the lower-bound launches of this run take the VL2 kernels)
};

// lanes per read of the latency tier (k_dp_skew W; IpxDims::lat): 32 -- four reads per wave, up to 8 segments = 256 rows
#define IPX_MAX_SEG 64       // largest segLen handled by the register-resident kernels
#define IPX_SLOW_BASE (IPX_MAX_SEG + 1)      // class of a job with gap_open <= gap_ext: segLen + IPX_SLOW_BASE
#define IPX_NUM_CLASSES (2 * IPX_SLOW_BASE)
#define IPX_MAX_REFLEN 32000 // columns are counted in 16-bit halves of packed registers (up to 32 767); column maxima in LDS when they fit, else global
#define IPX_REF_PAD 8        // window starts are 4-byte aligned, with >= 4 readable bytes after the end

// The batch as the kernels see it (all pointers are device pointers).
struct IpxBatch {
    int64_t n_jobs;
    int32_t n_refs;
    const int8_t *reads;        // concatenated read codes 0..4
    const int64_t *read_off;    // n_jobs+1
    const int8_t *refs_packed;  // windows re-packed: start 4-byte aligned, padded (k_pack_refs)
    const int64_t *refp_off;    // n_refs: byte offset of each packed window
    const int32_t *ref_len;     // n_refs
    const int32_t *ref_id;      // n_jobs
    const uint8_t *gap_open;    // n_jobs  (already narrowed to uint8, ssw.h:129-130)
    const uint8_t *gap_ext;     // n_jobs
    const int32_t *mask_len;    // n_jobs or nullptr -> max(15, readLen/2)  (sswpy.pyx:209-211)
    int8_t mat[25];             // 5x5 substitution matrix (sswpy.pyx:306-336)
    int32_t bias;               // |min(mat)| (ssw.c:795-799)
    int32_t word_first_len;     // reads at least this long take the 16-bit pass first (0 = never); speed only
    int32_t bracket_min_len;    // reads at least this long take the lower/upper-bound bracket when the lower-bound stage cannot
                                //   settle them, shorter ones (few columns with big carries) the stepped pass directly; speed only
    int32_t byte_safe_len;      // reads shorter than this cannot reach 255-bias (len * max(mat) < 255-bias): the lower-bound stage
                                //   has nothing to offer them but the bracket; below bracket_min_len they start in the stepped
                                //   pass directly; speed only
    int32_t f16_max_len;        // reads up to this length may take the half-precision form of the 16-bit passes (k_dp_pass F16): every
                                //   matrix entry is a half whose low byte is 0 and len * max(mat) <= 2047; 0 = never; speed only
    uint8_t use_bracket;        // an upper-bound stage exists for this batch (selector-profile kernels): speed only
    uint8_t plain_first;        // the 8-bit passes of this batch take the plain-first flow (IPX_PASS_BYTE_FIRST): speed only
    int32_t max_match;          // largest matrix entry
    int32_t lat_prio;           // the latency-bound kernels raise their waves' issue priority (s_setprio; off: IPX_ROUTE_NO_SETPRIO)
    int32_t plain_max_len;      // plain-first flow: reads up to this length take the plain kernels, longer 8-bit starters the stepped pass at once
    uint8_t exact_direct;       // what the proofs (k_prove_overflow, k_prove_plain) leave open goes to the stepped pass at once: speed only
    const uint8_t *cls_map;     // [IPX_NUM_PASSES][IPX_NUM_CLASSES] class a job of (pass, class) is LISTED under, or nullptr = its own.  The
                                //   wavefront kernels serve any read whose padded row count fits theirs (rows shifted down, k_dp_skew), so a
                                //   rare class rides in the next populated one's launch instead of getting a launch to itself: speed only
    uint8_t flag;               // ssw_align flag (ssw.c:821)
    uint8_t score_size;         // ssw_init score_size: 0 byte only, 1 word only, 2 both (ssw.c:793-802)
    uint16_t filters;
    int32_t filterd;
    IpxResult *res;             // n_jobs
    uint32_t *cigar_pool;
    uint32_t cigar_cap;         // capacity of cigar_pool in ops
    uint32_t *cigar_cursor;     // bump allocator (1 word)
    uint32_t *plan_counts;      // jobs per pass and class: row `pass` starts at plan_counts + pass * 2 * IPX_NUM_CLASSES (the pass's
                                //   scatter cursors follow its counts); the rows of the dynamic passes are zeroed at the start of a
                                //   run and filled by the kernels that decide a job's next pass
    uint32_t *maxcol_scratch;   // per DP block: column maxima of the tile in flight (forward passes)
    uint16_t *tb_bw;            // n_jobs: band half-width a later traceback tier starts the job from (0: its own first band, ssw.c:899), written by the
                                //   tier that hands the job over (k_tb_list zeroes it); only with tb_diag
    uint8_t tb_diag;            // the anti-diagonal traceback tiers (k_tb_diag) are in use: speed only
    uint32_t *status;           // bit0: cigar pool exhausted, bit1: read too long, bit2: ref too long, bit3: traceback scratch exhausted
};

// Planner output for one pass: jobs bucketed by class.
struct IpxPlan {
    uint32_t *count;      // [IPX_NUM_CLASSES] jobs per class (a row of IpxBatch::plan_counts)
    uint32_t *cursor;     // [IPX_NUM_CLASSES] scatter cursors
    uint32_t *cls_off;    // [IPX_NUM_CLASSES+1] first slot of class in perm
    uint32_t *tile_off;   // [IPX_NUM_CLASSES+1] first tile of class
    uint32_t *perm;       // [n_jobs] job ids grouped by class
    uint32_t *stats;      // [IPX_NUM_CLASSES+1] tiles per class of this pass's last plan (last entry: all classes); read back
                          //   by the host after a run to size the next run's launches (speed only); may be null
};

enum {
    IPX_STATUS_CIGAR_POOL = 1,
    IPX_STATUS_READ_TOO_LONG = 2,
    IPX_STATUS_REF_TOO_LONG = 4,
    IPX_STATUS_TB_SCRATCH = 8,
    IPX_STATUS_INTERNAL = 16,      // a kernel variant met a job it was not built for (host-side routing error)
    IPX_STATUS_RERUN = 32,         // a job was left in a pass the host had predicted empty and not launched (latency tier, ipx_run_pipeline `skip`): the run is repeated with every pass
};
