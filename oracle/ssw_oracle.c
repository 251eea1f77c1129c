/*
 * ssw_oracle.c -- CPU restatement of indelPost's striped Smith-Waterman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product path is the HIP library in
 * indelpost_amd/csrc and fails loudly when that is missing.
 *
 * Parity status: PINNED.  This restatement is checked (tests/test_oracle.py) against
 *   (1) the known-answer vectors of SURVEY.md section 8c, committed in tests/golden/, and
 *   (2) golden vectors generated in the build container by the reference's own ssw.c compiled
 *       unmodified from /root/reference into oracle/_ref/ (oracle/gen_golden.py), and, when
 *       oracle/_ref/libssw_ref.so is present, directly against that library on random cases.
 *
 * What is restated (all citations are into /root/reference/indelpost/):
 *   orc_striped_pass()  <- sw_sse2_byte  ssw.c:197-384   (lanes=16, unsigned 8-bit saturating)
 *                       <- sw_sse2_word  ssw.c:410-586   (lanes=8, signed 16-bit saturating add)
 *   profile values      <- qP_byte ssw.c:163-188, qP_word ssw.c:386-408 (looked up on the fly)
 *   orc_banded_path()   <- banded_sw     ssw.c:588-772
 *   orc_ssw_init()      <- ssw_init      ssw.c:787-808
 *   orc_ssw_align()     <- ssw_align     ssw.c:816-920
 *
 * The model is scalar: a "vector" of the reference is an array of `lanes` ints, the striped row
 * of (segment j, lane l) is r = j + l*segLen.  No SIMD, no attempt at speed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Same field order/types as s_align (ssw.h:55-66) so ctypes users can share one Structure. */
typedef struct {
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
} orc_align_t;

typedef struct {
    const int8_t *read; /* borrowed, like ssw.c:803 */
    const int8_t *mat;  /* borrowed, like ssw.c:804 */
    int32_t readLen;
    int32_t n;
    int32_t bias;
    int have_byte, have_word;
} orc_profile_t;

typedef struct {
    int score, ref, read;   /* best: score, 0-based end on ref, end on read */
    int score2, ref2;       /* second best outside the mask */
} orc_ends_t;

#define MAXLANES 16

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int subs0(int a, int b) { return a > b ? a - b : 0; } /* unsigned saturating a-b */

/*
 * One striped pass.  lanes==16: byte semantics (ssw.c:197-384); lanes==8: word semantics
 * (ssw.c:410-586).  dir: 0 walk ref left->right, 1 right->left (ssw.c:253-257 / 457-461).
 */
static orc_ends_t orc_striped_pass(const int8_t *ref, int dir, int refLen, const int8_t *read,
                                   int readLen, const int8_t *mat, int n, int gapO, int gapE,
                                   int lanes, int bias, int terminate, int maskLen)
{
    const int byte = (lanes == 16);
    const int segLen = (readLen + lanes - 1) / lanes;     /* ssw.c:221 / 428 */
    const int rows = segLen * lanes;
    int best = 0;
    int end_read = readLen - 1;                           /* ssw.c:219 / 426 */
    int end_ref = byte ? -1 : 0;                          /* ssw.c:220 / 427 */
    int *maxColumn = (int *)calloc(refLen > 0 ? refLen : 1, sizeof(int));
    int *Hst = (int *)calloc(rows > 0 ? rows : 1, sizeof(int)); /* pvHStore, index l*segLen+j */
    int *Hld = (int *)calloc(rows > 0 ? rows : 1, sizeof(int)); /* pvHLoad */
    int *E = (int *)calloc(rows > 0 ? rows : 1, sizeof(int));
    int *Hmax = (int *)calloc(rows > 0 ? rows : 1, sizeof(int));
    int vMaxScore[MAXLANES], vMaxMark[MAXLANES];
    int i, j, k, l;
    int begin = 0, end = refLen, step = 1;
    memset(vMaxScore, 0, sizeof vMaxScore);
    memset(vMaxMark, 0, sizeof vMaxMark);
    if (dir == 1) { begin = refLen - 1; end = -1; step = -1; }

    for (i = begin; i != end; i += step) {
        int vF[MAXLANES], vH[MAXLANES], colmax[MAXLANES];
        const int8_t *mrow = mat + (int)ref[i] * n;
        int changed, stop = 0, cm;
        int *tmp;
        for (l = 0; l < lanes; ++l) { vF[l] = 0; colmax[l] = 0; }
        /* vH = pvHStore[segLen-1] shifted one lane up, zero shifted in (ssw.c:264-265/467-468) */
        for (l = lanes - 1; l > 0; --l) vH[l] = segLen > 0 ? Hst[(l - 1) * segLen + segLen - 1] : 0;
        vH[0] = 0;
        tmp = Hld; Hld = Hst; Hst = tmp;                  /* swap (ssw.c:269-271/471-477) */

        for (j = 0; j < segLen; ++j) {                    /* ssw.c:274-299 / 480-504 */
            for (l = 0; l < lanes; ++l) {
                const int r = j + l * segLen;
                const int p = r < readLen ? mrow[read[r]] : 0; /* pad rows: bias / 0 */
                int h, t, e = E[l * segLen + j];
                if (byte) {
                    h = vH[l] + p + bias;                 /* adds_epu8 */
                    if (h > 255) h = 255;
                    h = subs0(h, bias);                   /* subs_epu8 vBias */
                } else {
                    h = vH[l] + p;                        /* adds_epi16 */
                    if (h > 32767) h = 32767;
                    if (h < -32768) h = -32768;
                }
                h = imax(h, e);
                h = imax(h, vF[l]);
                colmax[l] = imax(colmax[l], h);
                Hst[l * segLen + j] = h;
                t = subs0(h, gapO);
                e = subs0(e, gapE);
                E[l * segLen + j] = imax(e, t);
                vF[l] = imax(subs0(vF[l], gapE), t);
                vH[l] = Hld[l * segLen + j];
            }
        }

        /* Lazy-F (ssw.c:302-313 / 507-518): E is not corrected. */
        for (k = 0; k < lanes && !stop; ++k) {
            for (l = lanes - 1; l > 0; --l) vF[l] = vF[l - 1];
            vF[0] = 0;
            for (j = 0; j < segLen; ++j) {
                int any = 0;
                for (l = 0; l < lanes; ++l) {
                    int h = imax(Hst[l * segLen + j], vF[l]);
                    int f, hh;
                    colmax[l] = imax(colmax[l], h);
                    Hst[l * segLen + j] = h;
                    hh = subs0(h, gapO);
                    vF[l] = f = subs0(vF[l], gapE);
                    if (byte) {                           /* _mm_cmpgt_epi8: SIGNED bytes, ssw.c:311 */
                        if ((int8_t)(uint8_t)f > (int8_t)(uint8_t)hh) any = 1;
                    } else {                              /* _mm_cmpgt_epi16, ssw.c:516 */
                        if ((int16_t)(uint16_t)f > (int16_t)(uint16_t)hh) any = 1;
                    }
                }
                if (!any) { stop = 1; break; }
            }
        }

        /* running lane-wise maximum and the "did any lane change" test (ssw.c:316-333/521-535) */
        changed = 0;
        for (l = 0; l < lanes; ++l) {
            vMaxScore[l] = imax(vMaxScore[l], colmax[l]);
            if (vMaxScore[l] != vMaxMark[l]) changed = 1;
        }
        if (changed) {
            int temp = 0;
            for (l = 0; l < lanes; ++l) { vMaxMark[l] = vMaxScore[l]; temp = imax(temp, vMaxScore[l]); }
            if (temp > best) {
                best = temp;
                if (byte && best + bias >= 255) break;    /* overflow, ssw.c:327 */
                end_ref = i;
                memcpy(Hmax, Hst, sizeof(int) * (size_t)rows);
            }
        }
        cm = 0;
        for (l = 0; l < lanes; ++l) cm = imax(cm, colmax[l]);
        maxColumn[i] = cm;                                /* ssw.c:336 / 538 */
        if (cm == terminate) break;                       /* ssw.c:337 / 539 */
    }

    /* end position on the read: smallest row holding `best` in the saved column (ssw.c:340-349) */
    for (l = 0; l < lanes; ++l)
        for (j = 0; j < segLen; ++j)
            if (Hmax[l * segLen + j] == best) {
                int r = j + l * segLen;
                if (r < end_read) end_read = r;
            }

    orc_ends_t o;
    o.score = (byte && best + bias >= 255) ? 255 : best;  /* ssw.c:358 / 560 */
    o.ref = end_ref;
    o.read = end_read;
    o.score2 = 0;
    o.ref2 = 0;
    {   /* second best outside [end_ref-maskLen, end_ref+maskLen] (ssw.c:366-379 / 568-581) */
        int edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
        for (i = 0; i < edge; ++i)
            if (maxColumn[i] > o.score2) { o.score2 = maxColumn[i]; o.ref2 = i; }
        edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
        for (i = byte ? edge + 1 : edge; i < refLen; ++i)
            if (maxColumn[i] > o.score2) { o.score2 = maxColumn[i]; o.ref2 = i; }
    }
    free(maxColumn); free(Hst); free(Hld); free(E); free(Hmax);
    return o;
}

/* band-relative column of cell (i,j): ssw.c:92 (set_u) */
static inline int band_u(int w, int i, int j) { int x = i - w; if (x < 0) x = 0; return j - x + 1; }
/* band-relative direction slot: ssw.c:95 (set_d) */
static inline int band_d(int w, int i, int j, int p) { int x = i - w; if (x < 0) x = 0; return (j - x) * 3 + p; }

/*
 * Banded affine DP with traceback (banded_sw, ssw.c:588-772).  Returns BAM-encoded cigar
 * (len<<4|op, M=0 I=1 D=2) in *out (malloc'd) and its length; returns 0 on "trace back error".
 * Direction cells the reference would read uninitialised are zero here and take the error path.
 */
/* test visibility: first and final band width of the last traceback on this thread's caller (diagnostics only) */
int orc_dbg_band_first = 0, orc_dbg_band_final = 0;
static int orc_banded_path(const int8_t *ref, const int8_t *read, int refLen, int readLen, int score,
                           int gapO, int gapE, int band_width, const int8_t *mat, int n,
                           uint32_t **out, int *outLen)
{
    const int len = refLen > readLen ? refLen : readLen;
    int max = 0, width = 0, width_d = 0;
    int *h_b = NULL, *e_b = NULL, *h_c = NULL;
    int8_t *direction = NULL;
    size_t dir_cap = 0;
    int cap = 0;
    int i, j;

    orc_dbg_band_first = band_width;
    do {
        width = band_width * 2 + 3;
        width_d = band_width * 2 + 1;
        if (width + 1 > cap) {
            int newcap = width + 1, q;
            h_b = (int *)realloc(h_b, sizeof(int) * (size_t)newcap);
            e_b = (int *)realloc(e_b, sizeof(int) * (size_t)newcap);
            h_c = (int *)realloc(h_c, sizeof(int) * (size_t)newcap);
            for (q = cap; q < newcap; ++q) h_b[q] = e_b[q] = h_c[q] = 0;
            cap = newcap;
        }
        {
            size_t need = (size_t)width_d * (size_t)(readLen > 0 ? readLen : 1) * 3 + 16;
            if (need > dir_cap) {
                direction = (int8_t *)realloc(direction, need);
                memset(direction + dir_cap, 0, need - dir_cap);
                dir_cap = need;
            }
        }
        for (j = 1; j < width - 1; ++j) h_b[j] = 0;                    /* ssw.c:627 */
        for (i = 0; i < readLen; ++i) {
            int beg = 0, end = refLen - 1, u = 0, edge, f;
            int8_t *line = direction + (size_t)width_d * (size_t)i * 3;
            if (i - band_width > beg) beg = i - band_width;
            if (i + band_width < end) end = i + band_width;
            edge = end + 1 < width - 1 ? end + 1 : width - 1;          /* ssw.c:632 */
            f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;  /* ssw.c:633 */
            for (j = beg; j <= end; ++j) {
                int e, b, d, de, df, dh, t1, t2, e1, f1;
                u = band_u(band_width, i, j);
                e = band_u(band_width, i - 1, j);
                b = band_u(band_width, i, j - 1);
                d = band_u(band_width, i - 1, j - 1);
                de = band_d(band_width, i, j, 0);
                df = band_d(band_width, i, j, 1);
                dh = band_d(band_width, i, j, 2);

                t1 = i == 0 ? -gapO : h_b[e] - gapO;                   /* ssw.c:644-648 */
                t2 = i == 0 ? -gapE : e_b[e] - gapE;
                e_b[u] = t1 > t2 ? t1 : t2;
                line[de] = t1 > t2 ? 3 : 2;

                t1 = h_c[b] - gapO;                                    /* ssw.c:650-653 */
                t2 = f - gapE;
                f = t1 > t2 ? t1 : t2;
                line[df] = t1 > t2 ? 5 : 4;

                e1 = e_b[u] > 0 ? e_b[u] : 0;                          /* ssw.c:655-664 */
                f1 = f > 0 ? f : 0;
                t1 = e1 > f1 ? e1 : f1;
                t2 = h_b[d] + mat[(int)ref[j] * n + read[i]];
                h_c[u] = t1 > t2 ? t1 : t2;
                if (h_c[u] > max) max = h_c[u];
                if (t1 <= t2) line[dh] = 1;
                else line[dh] = e1 > f1 ? line[de] : line[df];
            }
            for (j = 1; j <= u; ++j) h_b[j] = h_c[j];                  /* ssw.c:666 */
        }
        band_width *= 2;
    } while (max < score && band_width <= len);                        /* ssw.c:669 */
    band_width /= 2;
    orc_dbg_band_final = band_width;

    {   /* trace back from the bottom-right cell (ssw.c:673-733) */
        int cap_c = 16, l = 0, e = 0, plane = 2;
        uint32_t *c = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap_c);
        char op = 'M', prev = 'M';
        int8_t *line = direction + (size_t)width_d * (size_t)(readLen > 0 ? readLen - 1 : 0) * 3;
        if (readLen <= 0) line = direction;
        i = readLen - 1;
        j = refLen - 1;
        while (i >= 0 && j > 0) {
            int code = line[band_d(band_width, i, j, plane)];
            switch (code) {
            case 1: --i; --j; plane = 2; line -= width_d * 3; op = 'M'; break;
            case 2: --i; plane = 0; line -= width_d * 3; op = 'I'; break;
            case 3: --i; plane = 2; line -= width_d * 3; op = 'I'; break;
            case 4: --j; plane = 1; op = 'D'; break;
            case 5: --j; plane = 2; op = 'D'; break;
            default:
                free(direction); free(h_b); free(e_b); free(h_c); free(c);
                *out = NULL; *outLen = 0;
                return 0;
            }
            if (op == prev) ++e;
            else {
                ++l;
                if (l + 2 >= cap_c) { cap_c *= 2; c = (uint32_t *)realloc(c, sizeof(uint32_t) * (size_t)cap_c); }
                c[l - 1] = ((uint32_t)e << 4) | (prev == 'M' ? 0u : prev == 'I' ? 1u : 2u);
                prev = op;
                e = 1;
            }
        }
        if (l + 3 >= cap_c) { cap_c += 4; c = (uint32_t *)realloc(c, sizeof(uint32_t) * (size_t)cap_c); }
        if (op == 'M') {                                               /* ssw.c:734-751 */
            ++l;
            c[l - 1] = ((uint32_t)(e + 1) << 4) | 0u;
        } else {
            l += 2;
            c[l - 2] = ((uint32_t)e << 4) | (op == 'I' ? 1u : 2u);
            c[l - 1] = (1u << 4) | 0u;
        }
        {   /* reverse into the output (ssw.c:754-762) */
            uint32_t *c1 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)l);
            int s;
            for (s = 0; s < l; ++s) c1[s] = c[l - 1 - s];
            *out = c1;
            *outLen = l;
        }
        free(c);
    }
    free(direction); free(h_b); free(e_b); free(h_c);
    return 1;
}

orc_profile_t *orc_ssw_init(const int8_t *read, int32_t readLen, const int8_t *mat, int32_t n,
                            int8_t score_size)
{
    orc_profile_t *p = (orc_profile_t *)calloc(1, sizeof *p);
    if (score_size == 0 || score_size == 2) {             /* ssw.c:793-801 */
        int bias = 0, i;
        for (i = 0; i < n * n; ++i) if (mat[i] < bias) bias = mat[i];
        p->bias = bias < 0 ? -bias : bias;
        p->have_byte = 1;
    }
    if (score_size == 1 || score_size == 2) p->have_word = 1;
    p->read = read; p->mat = mat; p->readLen = readLen; p->n = n;
    return p;
}

void orc_init_destroy(orc_profile_t *p) { free(p); }

orc_align_t *orc_ssw_align(const orc_profile_t *prof, const int8_t *ref, int32_t refLen,
                           uint8_t gapO, uint8_t gapE, uint8_t flag, uint16_t filters,
                           int32_t filterd, int32_t maskLen)
{
    orc_ends_t b, rv;
    int word = 0;
    int readLen = prof->readLen;
    orc_align_t *r = (orc_align_t *)calloc(1, sizeof *r);
    r->ref_begin1 = -1;
    r->read_begin1 = -1;

    if (prof->have_byte) {                                /* ssw.c:842-852 */
        b = orc_striped_pass(ref, 0, refLen, prof->read, readLen, prof->mat, prof->n, gapO, gapE, 16,
                             prof->bias, 255 /* (uint8_t)-1 */, maskLen);
        if (prof->have_word && b.score == 255) {
            b = orc_striped_pass(ref, 0, refLen, prof->read, readLen, prof->mat, prof->n, gapO, gapE,
                                 8, 0, 65535, maskLen);
            word = 1;
        } else if (b.score == 255) { free(r); return NULL; }
    } else if (prof->have_word) {
        b = orc_striped_pass(ref, 0, refLen, prof->read, readLen, prof->mat, prof->n, gapO, gapE, 8, 0,
                             65535, maskLen);
        word = 1;
    } else { free(r); return NULL; }

    r->score1 = (uint16_t)b.score;
    r->ref_end1 = b.ref;
    r->read_end1 = b.read;
    if (maskLen >= 15) { r->score2 = (uint16_t)b.score2; r->ref_end2 = b.ref2; }
    else { r->score2 = 0; r->ref_end2 = -1; }
    if (flag == 0 || (flag == 2 && r->score1 < filters)) return r;    /* ssw.c:872 */

    {   /* begin position: reversed read prefix vs ref prefix, right to left (ssw.c:875-886) */
        int n1 = r->read_end1 + 1, q;
        int8_t *rr = (int8_t *)calloc(n1 > 0 ? (size_t)n1 : 1, 1);
        for (q = 0; q < n1; ++q) rr[q] = prof->read[n1 - 1 - q];
        rv = orc_striped_pass(ref, 1, r->ref_end1 + 1, rr, n1, prof->mat, prof->n, gapO, gapE,
                              word ? 8 : 16, word ? 0 : prof->bias, r->score1, maskLen);
        free(rr);
    }
    r->ref_begin1 = rv.ref;
    r->read_begin1 = r->read_end1 - rv.read;
    if (r->score1 > rv.score) r->flag = 2;                            /* ssw.c:888-891 */

    if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
        ((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd ||
                             r->read_end1 - r->read_begin1 > filterd)))
        return r;                                                      /* ssw.c:894 */

    {   /* cigar (ssw.c:897-916) */
        int rl = r->ref_end1 - r->ref_begin1 + 1;
        int ql = r->read_end1 - r->read_begin1 + 1;
        int bw = abs(rl - ql) + 1;
        uint32_t *cg = NULL;
        int cl = 0, ok;
        const int8_t *rp = ref + r->ref_begin1;
        int8_t *guard = NULL;
        if (r->ref_begin1 < 0) {
            /* score-0 results start at ref[-1] in the reference (value never affects the path);
               read a zero there instead of out-of-bounds memory */
            int q;
            guard = (int8_t *)calloc((size_t)(rl > 0 ? rl : 1), 1);
            for (q = 0; q < rl; ++q) {
                int idx = r->ref_begin1 + q;
                guard[q] = (idx >= 0 && idx < refLen) ? ref[idx] : 0;
            }
            rp = guard;
        }
        ok = orc_banded_path(rp, prof->read + r->read_begin1, rl, ql, r->score1, gapO, gapE, bw,
                             prof->mat, prof->n, &cg, &cl);
        free(guard);
        if (!ok) r->flag = 1;
        else { r->cigar = cg; r->cigarLen = cl; }
    }
    return r;
}

void orc_align_destroy(orc_align_t *a) { if (a) { free(a->cigar); free(a); } }

/* Diagnostic export for tests: one striped pass, results in out[5] = score,ref,read,score2,ref2. */
void orc_pass(const int8_t *ref, int dir, int refLen, const int8_t *read, int readLen,
              const int8_t *mat, int n, int gapO, int gapE, int lanes, int bias, int terminate,
              int maskLen, int32_t *out)
{
    orc_ends_t o = orc_striped_pass(ref, dir, refLen, read, readLen, mat, n, gapO & 255, gapE & 255,
                                    lanes, bias, terminate, maskLen);
    out[0] = o.score; out[1] = o.ref; out[2] = o.read; out[3] = o.score2; out[4] = o.ref2;
}
