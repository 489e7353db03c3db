/*
 * walk_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, "port" kind).
 *
 * Plain-C restatement of the integer/float hot path of hwwang55/GraphGAN
 * (reference tree at /root/reference, never read at run time):
 *
 *   orc_build_trees      <- src/GraphGAN/graph_gan.py:84-108  (construct_trees)
 *   orc_walk_sample      <- src/GraphGAN/graph_gan.py:225-270 (sample), with
 *                           src/GraphGAN/generator.py:21      (all_score = E.E^T + b, bias per column)
 *                           src/utils.py:131-133              (softmax)
 *                           np.random.choice(p=...)           (graph_gan.py:262: first j with cdf_j > u)
 *   orc_pairs_from_path  <- src/GraphGAN/graph_gan.py:272-291 (get_node_pairs_from_path)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  The product (graphgan_amd/) never does.
 *
 * ARITHMETIC SPEC.  The reference's float arithmetic is
 * unpinned (TF1.8 matmul order, numpy SIMD exp, unseeded MT19937), so the
 * engine and this oracle share a *written* specification instead; this file
 * is an independent implementation of that text, not a copy of the kernel:
 *   S1 dot16:   rows are zero-padded to a multiple of 4 floats; float4 chunk c
 *               belongs to virtual lane t = c mod 16; a lane accumulates its
 *               chunks in increasing c with acc = fmaf(x_i, y_i, acc) over the
 *               4 elements in order, starting from 0.0f; the 16 lane sums are
 *               combined by an xor butterfly with offsets 8,4,2,1
 *               (v[t] = v[t] + v[t^off]); score = dot + bias[j] (one fp32 add).
 *   S2 exp:     x = s - max (fp32); x < -28 -> 0; else Cephes-style expf in
 *               pure fp32 with explicit fmaf (see orc_expf).
 *   S3 weight:  w_j = (uint64) trunc(e_j * 2^40);  W = sum w_j (exact integers).
 *   S4 uniform: Philox4x32-10, key = (seed_lo, seed_hi), counter =
 *               (hop, walk_in_root, root_id, stream); m = (x0>>5)<<26 | (x1>>6)
 *               (53 bits, numpy legacy random_sample layout); u = m / 2^53.
 *   S5 choice:  t = floor(m * W / 2^53) (128-bit product); pick the first j
 *               whose inclusive prefix C_j > t  ==  first j with C_j/W > u,
 *               i.e. searchsorted(cdf, u, side='right') of np.random.choice.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ S4 */
static inline void mulhilo32(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo32(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo32(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform numerator m (u = m / 2^53) for one hop */
uint64_t orc_uniform53(uint64_t seed, uint32_t stream, uint32_t root, uint32_t walk, uint32_t hop) {
    uint32_t ctr[4] = {hop, walk, root, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    return ((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6);
}

/* ------------------------------------------------------------------ S2 */
float orc_expf(float x) {
    if (x < -28.0f) return 0.0f;
    float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    int k = (int)kf; /* in [-41, 0] */
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)(k + 127) << 23;
    return p * sc.f;
}

/* ------------------------------------------------------------------ S1 */
float orc_dot16(const float *a, const float *b, int ld4 /* padded row length, multiple of 4 */) {
    float lane[16];
    int nchunk = ld4 / 4;
    for (int t = 0; t < 16; ++t) {
        float acc = 0.0f;
        for (int c = t; c < nchunk; c += 16) {
            const float *x = a + 4 * c, *y = b + 4 * c;
            acc = fmaf(x[0], y[0], acc);
            acc = fmaf(x[1], y[1], acc);
            acc = fmaf(x[2], y[2], acc);
            acc = fmaf(x[3], y[3], acc);
        }
        lane[t] = acc;
    }
    for (int off = 8; off >= 1; off >>= 1) {
        float nxt[16];
        for (int t = 0; t < 16; ++t) nxt[t] = lane[t] + lane[t ^ off];
        memcpy(lane, nxt, sizeof(lane));
    }
    return lane[0];
}

/* ------------------------------------------------------------------ S3 */
uint64_t orc_weight(float e) { return (uint64_t)(e * 1099511627776.0f); }

/* S5: index selection among k weights */
static int orc_choose(const uint64_t *w, int k, uint64_t m) {
    uint64_t W = 0;
    for (int j = 0; j < k; ++j) W += w[j];
    unsigned __int128 prod = (unsigned __int128)m * (unsigned __int128)W;
    uint64_t t = (uint64_t)(prod >> 53);
    uint64_t c = 0;
    for (int j = 0; j < k; ++j) {
        c += w[j];
        if (c > t) return j;
    }
    return k - 1; /* unreachable: t < W */
}

/* --------------------------------------------------- graph_gan.py:84-108 */
/* Tree CSR: for tree slot r, node v: nbr[nbr_base[r] + off[r*(n+1)+v] .. off[r*(n+1)+v+1])
 * = [father, child_0, child_1, ...] (root: [root, child...]); lists stored in node-id order.
 * nbr_base must hold n_roots+1 entries; returns total entries, or -1 if cap too small
 * (nbr may be NULL to size only).  depth_max_out (optional): deepest BFS level seen. */
int64_t orc_build_trees(int n, const int64_t *rowptr, const int32_t *col,
                        const int32_t *roots, int n_roots,
                        int32_t *off, int32_t *nbr, int64_t *nbr_base, int64_t cap,
                        int32_t *depth_max_out) {
    int32_t *father = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *depth = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * n);
    uint8_t *used = (uint8_t *)malloc(n);
    int64_t total = 0;
    int32_t dmax = 0;
    for (int r = 0; r < n_roots; ++r) {
        int root = roots[r];
        memset(used, 0, n);
        memset(cnt, 0, sizeof(int32_t) * (n + 1));
        int qh = 0, qt = 0;
        queue[qt++] = root;
        used[root] = 1;
        father[root] = root;
        depth[root] = 0;
        cnt[root] = 1; /* the father slot */
        /* pass 1: BFS in reference pop order (FIFO, adjacency order) */
        while (qh < qt) {
            int cur = queue[qh++];
            for (int64_t e = rowptr[cur]; e < rowptr[cur + 1]; ++e) {
                int sub = col[e];
                if (!used[sub]) {
                    used[sub] = 1;
                    father[sub] = cur;
                    depth[sub] = depth[cur] + 1;
                    if (depth[sub] > dmax) dmax = depth[sub];
                    cnt[cur] += 1;
                    cnt[sub] = 1;
                    queue[qt++] = sub;
                }
            }
        }
        int32_t *o = off + (int64_t)r * (n + 1);
        int32_t run = 0;
        for (int v = 0; v < n; ++v) { o[v] = run; run += cnt[v]; }
        o[n] = run;
        nbr_base[r] = total;
        if (nbr) {
            if (total + run > cap) { total = -1; goto done; }
            int32_t *nb = nbr + total;
            for (int i = 0; i < qt; ++i) { int v = queue[i]; nb[o[v]] = father[v]; fill[v] = 1; }
            /* children are appended in discovery order == queue order restricted to a father */
            for (int i = 1; i < qt; ++i) {
                int v = queue[i], f = father[v];
                nb[o[f] + fill[f]] = v;
                fill[f] += 1;
            }
        }
        total += run;
    }
    nbr_base[n_roots] = total;
done:
    if (depth_max_out) *depth_max_out = dmax;
    free(father); free(queue); free(depth); free(cnt); free(fill); free(used);
    return total;
}

/* -------------------------------------------------- graph_gan.py:225-270 */
typedef struct {
    int64_t hops;      /* softmax-samples executed (one appended path element each) */
    int64_t nbr_reads; /* sum over hops of k (neighbour scores computed) */
    int64_t walks;     /* walks completed (not aborted) */
} orc_counters;

/*
 * Sequential walk sampler over the tree slots slots[0..n_slots), n_walks[i] walks each
 * (D-mode: deg(root) incl. self-loop duplicates, graph_gan.py:190-191; G-mode:
 * config.n_sample_gen, graph_gan.py:210).  Reproduces: root-only-children at hop 0
 * (:250), whole-root abort (:252-257), IN-PLACE removal of the root from a depth-1
 * child's list in D-mode (:258-259, persists in nbr -> visible to later calls),
 * stop when next == previous (:264-266).
 *   samples[w]         walk end node, -1 if the root aborted
 *   paths[w*stride..]  [root, ..., end, prev-of-end]; path_len[w] elements (0 if aborted)
 *   root_status[i]     0 ok, 1 aborted (reference returns (None, None)), 2 no walks requested
 * Returns 0, or -2 if a path would exceed stride.
 */
int orc_walk_sample(int n, int ld4, const float *E, const float *bias,
                    const int32_t *off, int32_t *nbr, const int64_t *nbr_base,
                    const int32_t *tree_root,
                    const int32_t *slots, const int32_t *n_walks, int n_slots, int for_d,
                    uint64_t seed, uint32_t stream,
                    int32_t *samples, int32_t *paths, int32_t *path_len, int stride,
                    int32_t *root_status, orc_counters *ctr) {
    int kcap = 1024;
    float *sc = (float *)malloc(sizeof(float) * kcap);
    uint64_t *w = (uint64_t *)malloc(sizeof(uint64_t) * kcap);
    int64_t wbase = 0;
    int rc = 0;
    if (ctr) memset(ctr, 0, sizeof(*ctr));
    for (int i = 0; i < n_slots; ++i) {
        int slot = slots[i];
        int root = tree_root[slot];
        const int32_t *o = off + (int64_t)slot * (n + 1);
        int32_t *nb = nbr + nbr_base[slot];
        int nw = n_walks[i];
        int aborted = 0;
        for (int j = 0; j < nw && !aborted; ++j) {
            int32_t *path = paths + (wbase + j) * (int64_t)stride;
            int cur = root, prev = -1, is_root = 1, len = 0;
            uint32_t hop = 0;
            path[len++] = cur;
            for (;;) {
                int beg = o[cur], end = o[cur + 1];
                if (is_root) beg += 1;               /* tree[cur][1:] */
                else if (nb[beg] < 0) beg += 1;      /* father entry already removed (Q3) */
                is_root = 0;
                int k = end - beg;
                if (k == 0) { aborted = 1; break; }  /* the tree only has a root */
                if (for_d) {
                    if (k == 1 && nb[beg] == root) { aborted = 1; break; }
                    if (nb[beg] == root) {           /* root can only sit in the father slot */
                        nb[beg] = -1;                /* node_neighbor.remove(root), in place */
                        beg += 1; k -= 1;
                    }
                }
                if (k > kcap) {
                    kcap = k * 2;
                    sc = (float *)realloc(sc, sizeof(float) * kcap);
                    w = (uint64_t *)realloc(w, sizeof(uint64_t) * kcap);
                }
                const float *gc = E + (int64_t)cur * ld4;
                float mx = -INFINITY;
                for (int q = 0; q < k; ++q) {
                    int v = nb[beg + q];
                    float s = orc_dot16(gc, E + (int64_t)v * ld4, ld4) + bias[v];
                    sc[q] = s;
                    if (s > mx) mx = s;
                }
                for (int q = 0; q < k; ++q) w[q] = orc_weight(orc_expf(sc[q] - mx));
                uint64_t m = orc_uniform53(seed, stream, (uint32_t)root, (uint32_t)j, hop);
                int nxt = nb[beg + orc_choose(w, k, m)];
                if (ctr) { ctr->hops += 1; ctr->nbr_reads += k; }
                if (len >= stride) { rc = -2; goto out; }
                path[len++] = nxt;
                hop += 1;
                if (nxt == prev) { samples[wbase + j] = cur; break; }
                prev = cur; cur = nxt;
            }
            path_len[wbase + j] = len;
        }
        if (aborted) {
            for (int j = 0; j < nw; ++j) { samples[wbase + j] = -1; path_len[wbase + j] = 0; }
            root_status[i] = 1;
        } else {
            root_status[i] = nw == 0 ? 2 : 0;
            if (ctr) ctr->walks += nw;
        }
        wbase += nw;
    }
out:
    free(sc); free(w);
    return rc;
}

/* -------------------------------------------------- graph_gan.py:272-291 */
/* path has len elements; the last one is dropped (:282); returns the number of pairs
 * written to (a[], b[]) (caller provides >= 2*window*len entries). */
int orc_pairs_from_path(const int32_t *path, int len, int window, int32_t *a, int32_t *b) {
    int L = len - 1, np = 0;
    for (int i = 0; i < L; ++i) {
        int lo = i - window < 0 ? 0 : i - window;
        int hi = i + window + 1 > L ? L : i + window + 1;
        for (int j = lo; j < hi; ++j) {
            if (j == i) continue;
            a[np] = path[i]; b[np] = path[j]; ++np;
        }
    }
    return np;
}

/* --------------------------------------------------------- generator.py:21 */
/* Rows of all_score = E.E^T + b (bias per column).  Arithmetic restated for the MFMA kernel
 * (v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fmaf chain from 0): acc = fmaf(a_k, b_k, acc) for
 * k = 0..ld-1 in order, then one fp32 add of the bias. */
void orc_all_score_rows(const float *E, const float *bias, int n, int ld, const int32_t *rows, int n_rows, float *out) {
    for (int i = 0; i < n_rows; ++i) {
        const float *a = E + (int64_t)(rows ? rows[i] : i) * ld;
        for (int j = 0; j < n; ++j) {
            const float *b = E + (int64_t)j * ld;
            float acc = 0.0f;
            for (int k = 0; k < ld; ++k) acc = fmaf(a[k], b[k], acc);
            out[(int64_t)i * n + j] = acc + bias[j];
        }
    }
}
