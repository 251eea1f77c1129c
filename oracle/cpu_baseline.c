/*
 * cpu_baseline.c -- threaded timing harness for the CPU leg of bench.py (TEST INFRASTRUCTURE).
 *
 * dlopen()s a library exporting the reference's four entry points (ssw.h:86,91,126-134,139) --
 * either oracle/_ref/libssw_ref.so (the reference's ssw.c compiled unmodified; prefix "") or this
 * directory's restatement (prefix "orc_") -- and runs the reference's per-read loop
 *   ssw_init(read,len,mat,5,2) -> ssw_align(prof,ref,refLen,go,ge,flag=1,0,0,max(15,len/2))
 *   -> align_destroy -> init_destroy                       (sswpy.pyx:172-177, 209-219)
 * over a static contiguous partition of the jobs on `nthreads` pthreads.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
    uint16_t score1, score2;
    int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
    uint32_t *cigar;
    int32_t cigarLen;
    uint16_t flag;
} sal_t;

typedef void *(*init_fn)(const int8_t *, int32_t, const int8_t *, int32_t, int8_t);
typedef sal_t *(*align_fn)(const void *, const int8_t *, int32_t, uint8_t, uint8_t, uint8_t, uint16_t,
                           int32_t, int32_t);
typedef void (*adestroy_fn)(sal_t *);
typedef void (*idestroy_fn)(void *);

typedef struct {
    init_fn init; align_fn align; adestroy_fn adestroy; idestroy_fn idestroy;
    const int8_t *reads; const int64_t *read_off;
    const int8_t *refs; const int64_t *ref_off; const int32_t *ref_id;
    const uint8_t *gapO, *gapE;
    const int8_t *mat;
    int64_t lo, hi;
    int64_t checksum, cigar_ops;
    int cpu;                 /* logical CPU this thread pins itself to, or -1 */
} work_t;

/* one pinned thread per physical core.  The caller (bench.py) picks the logical CPUs. */
static void pin_self(int cpu)
{
    if (cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);   /* best effort: a refused mask leaves the thread unpinned */
    }
}

static void *worker(void *arg)
{
    work_t *w = (work_t *)arg;
    int64_t k;
    pin_self(w->cpu);
    for (k = w->lo; k < w->hi; ++k) {
        const int8_t *rd = w->reads + w->read_off[k];
        int32_t rl = (int32_t)(w->read_off[k + 1] - w->read_off[k]);
        int32_t rid = w->ref_id[k];
        const int8_t *rf = w->refs + w->ref_off[rid];
        int32_t fl = (int32_t)(w->ref_off[rid + 1] - w->ref_off[rid]);
        int32_t mask = rl / 2 < 15 ? 15 : rl / 2;
        void *p = w->init(rd, rl, w->mat, 5, 2);
        sal_t *a = w->align(p, rf, fl, w->gapO[k], w->gapE[k], 1, 0, 0, mask);
        if (a) {
            w->checksum += a->score1;
            w->cigar_ops += a->cigarLen;
            w->adestroy(a);
        }
        w->idestroy(p);
    }
    return NULL;
}

/* returns 0 on success; seconds_out = wall time of the threaded section */
int ipx_cpu_baseline_pinned(const char *libpath, const char *prefix, const int8_t *reads,
                     const int64_t *read_off, const int8_t *refs, const int64_t *ref_off,
                     const int32_t *ref_id, const uint8_t *gapO, const uint8_t *gapE,
                     const int8_t *mat, int64_t n_jobs, int nthreads, const int32_t *cpus /* nthreads ids or NULL */,
                     double *seconds_out, int64_t *checksum_out, int64_t *cigar_ops_out)
{
    char name[128];
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    pthread_t *th;
    work_t *ws;
    struct timespec t0, t1;
    int t;
    if (!h) { fprintf(stderr, "cpu_baseline: dlopen %s: %s\n", libpath, dlerror()); return -1; }
    if (nthreads < 1) nthreads = 1;
    th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    ws = (work_t *)calloc((size_t)nthreads, sizeof *ws);
    for (t = 0; t < nthreads; ++t) {
        work_t *w = &ws[t];
        snprintf(name, sizeof name, "%sssw_init", prefix);      w->init = (init_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%sssw_align", prefix);     w->align = (align_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%salign_destroy", prefix); w->adestroy = (adestroy_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%sinit_destroy", prefix);  w->idestroy = (idestroy_fn)dlsym(h, name);
        if (!w->init || !w->align || !w->adestroy || !w->idestroy) {
            fprintf(stderr, "cpu_baseline: missing %sssw_* symbols in %s\n", prefix, libpath);
            free(th); free(ws); dlclose(h);
            return -2;
        }
        w->reads = reads; w->read_off = read_off; w->refs = refs; w->ref_off = ref_off;
        w->ref_id = ref_id; w->gapO = gapO; w->gapE = gapE; w->mat = mat;
        w->lo = n_jobs * t / nthreads;
        w->hi = n_jobs * (t + 1) / nthreads;
        w->cpu = cpus ? cpus[t] : -1;
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, &ws[t]);
    for (t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    *checksum_out = 0;
    *cigar_ops_out = 0;
    for (t = 0; t < nthreads; ++t) { *checksum_out += ws[t].checksum; *cigar_ops_out += ws[t].cigar_ops; }
    free(th); free(ws); dlclose(h);
    return 0;
}

int ipx_cpu_baseline(const char *libpath, const char *prefix, const int8_t *reads,
                     const int64_t *read_off, const int8_t *refs, const int64_t *ref_off,
                     const int32_t *ref_id, const uint8_t *gapO, const uint8_t *gapE,
                     const int8_t *mat, int64_t n_jobs, int nthreads, double *seconds_out,
                     int64_t *checksum_out, int64_t *cigar_ops_out)
{
    return ipx_cpu_baseline_pinned(libpath, prefix, reads, read_off, refs, ref_off, ref_id, gapO, gapE, mat, n_jobs,
                                   nthreads, NULL, seconds_out, checksum_out, cigar_ops_out);
}

/* ---- full per-job results from the CPU checker (parity stress tests) -------------------------------
 * Same threading as above; writes one 32-byte record per job with the layout of ipx_result
 * (include/indelpost_hip.h) except that cigar_off carries an FNV-1a hash of the BAM-encoded ops. */
typedef struct {
    uint16_t score1, score2;
    int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
    uint32_t cigar_hash;
    uint16_t cigar_len;
    uint8_t flag, is_null;
} rec_t;

typedef struct { work_t w; rec_t *out; uint32_t *wsum; /* optional: sum_q cigar[q]*(q+1) per job */ } work2_t;

static void *worker2(void *arg)
{
    work2_t *x = (work2_t *)arg;
    work_t *w = &x->w;
    int64_t k;
    for (k = w->lo; k < w->hi; ++k) {
        const int8_t *rd = w->reads + w->read_off[k];
        int32_t rl = (int32_t)(w->read_off[k + 1] - w->read_off[k]);
        int32_t rid = w->ref_id[k];
        const int8_t *rf = w->refs + w->ref_off[rid];
        int32_t fl = (int32_t)(w->ref_off[rid + 1] - w->ref_off[rid]);
        int32_t mask = rl / 2 < 15 ? 15 : rl / 2;
        void *p = w->init(rd, rl, w->mat, 5, 2);
        sal_t *a = w->align(p, rf, fl, w->gapO[k], w->gapE[k], 1, 0, 0, mask);
        rec_t *r = &x->out[k];
        r->is_null = a ? 0 : 1;
        if (a) {
            uint32_t h = 2166136261u;
            int q;
            r->score1 = a->score1; r->score2 = a->score2; r->ref_begin1 = a->ref_begin1; r->ref_end1 = a->ref_end1;
            r->read_begin1 = a->read_begin1; r->read_end1 = a->read_end1; r->ref_end2 = a->ref_end2;
            r->flag = (uint8_t)a->flag;
            r->cigar_len = (uint16_t)(a->cigar ? a->cigarLen : 0);
            for (q = 0; a->cigar && q < a->cigarLen; ++q) { h ^= a->cigar[q]; h *= 16777619u; }
            r->cigar_hash = h;
            if (x->wsum) {
                uint32_t ws = 0;
                for (q = 0; a->cigar && q < a->cigarLen; ++q) ws += a->cigar[q] * (uint32_t)(q + 1);
                x->wsum[k] = ws;
            }
            w->adestroy(a);
        }
        w->idestroy(p);
    }
    return NULL;
}

int ipx_cpu_batch_results_w(const char *libpath, const char *prefix, const int8_t *reads,
                          const int64_t *read_off, const int8_t *refs, const int64_t *ref_off,
                          const int32_t *ref_id, const uint8_t *gapO, const uint8_t *gapE,
                          const int8_t *mat, int64_t n_jobs, int nthreads, void *out_records, uint32_t *out_wsum)
{
    char name[128];
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    pthread_t *th;
    work2_t *ws;
    int t;
    if (!h) { fprintf(stderr, "cpu_batch: dlopen %s: %s\n", libpath, dlerror()); return -1; }
    if (nthreads < 1) nthreads = 1;
    th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    ws = (work2_t *)calloc((size_t)nthreads, sizeof *ws);
    for (t = 0; t < nthreads; ++t) {
        work_t *w = &ws[t].w;
        snprintf(name, sizeof name, "%sssw_init", prefix);      w->init = (init_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%sssw_align", prefix);     w->align = (align_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%salign_destroy", prefix); w->adestroy = (adestroy_fn)dlsym(h, name);
        snprintf(name, sizeof name, "%sinit_destroy", prefix);  w->idestroy = (idestroy_fn)dlsym(h, name);
        if (!w->init || !w->align || !w->adestroy || !w->idestroy) { free(th); free(ws); dlclose(h); return -2; }
        w->reads = reads; w->read_off = read_off; w->refs = refs; w->ref_off = ref_off;
        w->ref_id = ref_id; w->gapO = gapO; w->gapE = gapE; w->mat = mat;
        w->lo = n_jobs * t / nthreads;
        w->hi = n_jobs * (t + 1) / nthreads;
        w->cpu = -1;
        ws[t].out = (rec_t *)out_records;
        ws[t].wsum = out_wsum;
    }
    for (t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker2, &ws[t]);
    for (t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(ws); dlclose(h);
    return 0;
}

int ipx_cpu_batch_results(const char *libpath, const char *prefix, const int8_t *reads,
                          const int64_t *read_off, const int8_t *refs, const int64_t *ref_off,
                          const int32_t *ref_id, const uint8_t *gapO, const uint8_t *gapE,
                          const int8_t *mat, int64_t n_jobs, int nthreads, void *out_records)
{
    return ipx_cpu_batch_results_w(libpath, prefix, reads, read_off, refs, ref_off, ref_id, gapO, gapE, mat, n_jobs,
                                   nthreads, out_records, NULL);
}
