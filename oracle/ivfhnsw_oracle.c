/*
 * ivfhnsw_oracle.c -- CPU restatement of the uniio/ivf-hnsw IVFADC search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ivfhnsw_oracle.h).  PARITY UNPINNED: the reference ships no golden
 * vectors for this path and cannot be built in this image (faiss submodule empty); every function
 * below restates the cited reference lines, faiss leafs restate faiss's published algorithms.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: the float evaluation order written here IS the
 * contract; the compiler must not fuse or reassociate it).
 */
#include "ivfhnsw_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 0.00001 /* utils.h:31, a double literal: comparisons against it are done in double */

/* =============================================================================================
 * leaf arithmetic
 * ============================================================================================= */

/* hnswlib/hnswalg.cpp:326-357 (fstdistfunc) and utils.cpp:22-52 (fvec_L2sqr), AVX branch.
 * One __m256 accumulator = 8 lanes; each loop iteration consumes 16 floats as two 8-wide
 * sub/mul/add steps; TmpRes[0]+...+TmpRes[7] is summed left to right. */
#ifndef ORC_OFAST_ORDER
float orc_l2sqr(const float *x, const float *y, size_t d)
{
    /* one 8-wide vector accumulator == the 8 lanes; element-wise sub, mul, add, each rounded
     * (built with -ffp-contract=off, so no step is fused) */
    typedef float v8f __attribute__((vector_size(32), aligned(4)));
    v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t nblk = d >> 4;
    for (size_t b = 0; b < nblk; b++) {
        for (int half = 0; half < 2; half++) {
            v8f xv = *(const v8f *)(x + b * 16 + half * 8);
            v8f yv = *(const v8f *)(y + b * 16 + half * 8);
            v8f diff = xv - yv;
            v8f sq = diff * diff;
            acc = acc + sq;
        }
    }
    float res = acc[0] + acc[1];
    res = res + acc[2];
    res = res + acc[3];
    res = res + acc[4];
    res = res + acc[5];
    res = res + acc[6];
    res = res + acc[7];
    return res;
}
#else
/* The association g++ 11.4 ACTUALLY emits for that source under the reference's own flags (CMakeLists.txt:22,
 * -Ofast -march=native on an FMA machine), read from the assembly of the same loop shape:
 *   loop, per 16 floats:  t = d1*d1;  u = fma(d0, d0, t);  sum = sum + u     (vmulps, vfmadd132ps, vaddps)
 *   horizontal sum:       ((s5+s6) + (s3+s4)) + ((s7+s1) + (s0+s2))
 * Built as liborc_ofast.so (make -C oracle ofast) to MEASURE how often the two orders disagree on a top-1 id; the
 * source order above stays the contract. */
float orc_l2sqr(const float *x, const float *y, size_t d)
{
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t nblk = d >> 4;
    for (size_t b = 0; b < nblk; b++)
        for (int l = 0; l < 8; l++) {
            float d0 = x[b * 16 + l] - y[b * 16 + l];
            float d1 = x[b * 16 + 8 + l] - y[b * 16 + 8 + l];
            float t = d1 * d1;
            float u = fmaf(d0, d0, t);
            s[l] = s[l] + u;
        }
    float a = (s[5] + s[6]) + (s[3] + s[4]);
    float c = (s[7] + s[1]) + (s[0] + s[2]);
    return a + c;
}
#endif

/* faiss spec (utils.cpp, SSE build per CMakeLists.txt.faiss:24): fvec_inner_product keeps one
 * __m128 of 4 partial sums, consumes 4 floats per step, loads the tail zero padded, then
 * hadd(hadd(s)) = (s0+s1)+(s2+s3). */
static float orc_inner_product_sse_order(const float *x, const float *y, size_t n)
{
    float s[4] = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int l = 0; l < 4; l++) {
            float p = x[i + l] * y[i + l];
            s[l] = s[l] + p;
        }
    for (int l = 0; i + l < n; l++) { /* masked tail: missing lanes add +0 */
        float p = x[i + l] * y[i + l];
        s[l] = s[l] + p;
    }
    float a = s[0] + s[1];
    float b = s[2] + s[3];
    return a + b;
}

/* faiss spec: ProductQuantizer::compute_inner_prod_table -> fvec_inner_products_ny per sub-space.
 * Reference call sites IndexIVF_HNSW.cpp:262, IndexIVF_HNSW_Grouping.cpp:265. */
void orc_inner_prod_table(const float *x, const float *pq_centroids, size_t d, size_t M, float *tab)
{
    size_t dsub = d / M;
    for (size_t m = 0; m < M; m++)
        for (size_t c = 0; c < 256; c++)
            tab[m * 256 + c] =
                orc_inner_product_sse_order(x + m * dsub, pq_centroids + (m * 256 + c) * dsub, dsub);
}

/* faiss spec: LinearTransform::apply(1, x) for OPQ (no bias), call sites IndexIVF_HNSW.cpp:240,
 * IndexIVF_HNSW_Grouping.cpp:199.  BLAS summation order is unspecified; the contract chosen here is
 * the k-ordered fmaf chain (one rounding per step). */
void orc_opq_apply(const float *A, const float *x, size_t d, float *y)
{
    for (size_t i = 0; i < d; i++) {
        float acc = 0.0f;
        for (size_t k = 0; k < d; k++)
            acc = fmaf(A[i * d + k], x[k], acc);
        y[i] = acc;
    }
}

/* faiss spec: Heap.h heap_heapify<CMax<float,long>> with no input: neutral = FLT_MAX, id = -1. */
void orc_maxheap_heapify(size_t k, float *val, long *ids)
{
    for (size_t i = 0; i < k; i++) {
        val[i] = FLT_MAX;
        ids[i] = -1;
    }
}

/* faiss spec: Heap.h heap_pop<CMax>: 1-based sift-down of the last element from the root;
 * comparator cmp(a,b) = a > b, values only. */
void orc_maxheap_pop(size_t k, float *val, long *ids)
{
    float *v = val - 1;
    long *id = ids - 1;
    float last = v[k];
    size_t i = 1;
    for (;;) {
        size_t l = i << 1, r = l + 1;
        if (l > k)
            break;
        if (r == k + 1 || v[l] > v[r]) {
            if (last > v[l])
                break;
            v[i] = v[l];
            id[i] = id[l];
            i = l;
        } else {
            if (last > v[r])
                break;
            v[i] = v[r];
            id[i] = id[r];
            i = r;
        }
    }
    v[i] = v[k];
    id[i] = id[k];
}

/* faiss spec: Heap.h heap_push<CMax>: 1-based sift-up from slot k. */
void orc_maxheap_push(size_t k, float *val, long *ids, float nv, long nid)
{
    float *v = val - 1;
    long *id = ids - 1;
    size_t i = k;
    while (i > 1) {
        size_t f = i >> 1;
        if (!(nv > v[f]))
            break;
        v[i] = v[f];
        id[i] = id[f];
        i = f;
    }
    v[i] = nv;
    id[i] = nid;
}

/* =============================================================================================
 * std::priority_queue<std::pair<float, idx_t>> stand-in: binary max-heap under pair's operator<
 * (first, then second).  Any correct heap pops the same sequence because ids are unique.
 * ============================================================================================= */
typedef struct { float key; uint32_t id; } orc_pair;
typedef struct { orc_pair *a; size_t n, cap; } orc_pq;

static int pair_less(orc_pair x, orc_pair y)
{
    return x.key < y.key || (!(y.key < x.key) && x.id < y.id);
}
static void pq_init(orc_pq *q) { q->a = NULL; q->n = q->cap = 0; }
static void pq_free(orc_pq *q) { free(q->a); q->a = NULL; q->n = q->cap = 0; }
static void pq_push(orc_pq *q, float key, uint32_t id)
{
    if (q->n == q->cap) {
        q->cap = q->cap ? q->cap * 2 : 64;
        q->a = (orc_pair *)realloc(q->a, q->cap * sizeof(orc_pair));
    }
    size_t i = q->n++;
    orc_pair p = {key, id};
    while (i > 0) {
        size_t f = (i - 1) >> 1;
        if (!pair_less(q->a[f], p))
            break;
        q->a[i] = q->a[f];
        i = f;
    }
    q->a[i] = p;
}
static orc_pair pq_top(const orc_pq *q) { return q->a[0]; }
static void pq_pop(orc_pq *q)
{
    orc_pair p = q->a[--q->n];
    size_t i = 0, n = q->n;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m;
        if (l >= n)
            break;
        m = (r < n && pair_less(q->a[l], q->a[r])) ? r : l;
        if (!pair_less(p, q->a[m]))
            break;
        q->a[i] = q->a[m];
        i = m;
    }
    if (n)
        q->a[i] = p;
}

/* =============================================================================================
 * HNSW coarse quantizer -- hnswlib/hnswalg.{h,cpp}
 * ============================================================================================= */
struct orc_hnsw {
    size_t d, n, maxelements, M, maxM, efConstruction;
    uint32_t enterpoint;
    uint8_t *counts;  /* the 1-byte link count of hnswalg.cpp:25 */
    uint32_t *links;  /* maxM slots per node */
    float *vectors;
    uint16_t *visited; /* visited_list_pool.h:8-33 */
    uint16_t epoch;
    unsigned long long dist_calc;
};

orc_hnsw *orc_hnsw_new(size_t d, size_t maxelements, size_t M, size_t maxM, size_t efConstruction)
{
    orc_hnsw *g = (orc_hnsw *)calloc(1, sizeof(*g));
    g->d = d;
    g->maxelements = maxelements;
    g->M = M;
    g->maxM = maxM;
    g->efConstruction = efConstruction;
    g->enterpoint = 0; /* hnswalg.cpp:37 */
    g->counts = (uint8_t *)calloc(maxelements, 1);
    g->links = (uint32_t *)calloc(maxelements * maxM, sizeof(uint32_t));
    g->vectors = (float *)calloc(maxelements * d, sizeof(float));
    g->visited = (uint16_t *)calloc(maxelements, sizeof(uint16_t));
    g->epoch = (uint16_t)-1; /* visited_list_pool.h:17 */
    return g;
}

void orc_hnsw_free(orc_hnsw *g)
{
    if (!g)
        return;
    free(g->counts);
    free(g->links);
    free(g->vectors);
    free(g->visited);
    free(g);
}

orc_hnsw *orc_hnsw_from_arrays(size_t d, size_t n, size_t M, size_t maxM, uint32_t enterpoint,
                               const uint8_t *counts, const uint32_t *links, const float *vectors)
{
    orc_hnsw *g = orc_hnsw_new(d, n, M, maxM, 0);
    g->n = n;
    g->enterpoint = enterpoint;
    memcpy(g->counts, counts, n);
    memcpy(g->links, links, n * maxM * sizeof(uint32_t));
    memcpy(g->vectors, vectors, n * d * sizeof(float));
    return g;
}

size_t orc_hnsw_n(const orc_hnsw *g) { return g->n; }
size_t orc_hnsw_d(const orc_hnsw *g) { return g->d; }
size_t orc_hnsw_maxM(const orc_hnsw *g) { return g->maxM; }
uint32_t orc_hnsw_enterpoint(const orc_hnsw *g) { return g->enterpoint; }
const uint8_t *orc_hnsw_counts(const orc_hnsw *g) { return g->counts; }
const uint32_t *orc_hnsw_links(const orc_hnsw *g) { return g->links; }
float *orc_hnsw_vectors(orc_hnsw *g) { return g->vectors; }
unsigned long long orc_hnsw_dist_calc(const orc_hnsw *g) { return g->dist_calc; }

/* visited_list_pool.h:25-32 reset(): epoch counter with wrap-around clear. */
static uint16_t visited_next_epoch(uint16_t *mass, uint16_t *cur, size_t n)
{
    (*cur)++;
    if (*cur == 0) {
        memset(mass, 0, sizeof(uint16_t) * n);
        (*cur)++;
    }
    return *cur;
}

/* hnswalg.cpp:48-109 searchBaseLayer.  `top` receives the result heap (max-heap of <= ef pairs).
 * visited/epoch are passed in so that the batch driver can use per-thread lists
 * (visited_list_pool.h:55-76 hands every concurrent caller its own list). */
static void hnsw_search_base_layer(const orc_hnsw *g, const float *point, size_t ef, orc_pq *top,
                                   uint16_t *mass, uint16_t *cur_epoch, unsigned long long *dist_calc)
{
    orc_pq cand;
    pq_init(&cand);
    uint16_t cv = visited_next_epoch(mass, cur_epoch, g->maxelements);

    float dist = orc_l2sqr(point, g->vectors + (size_t)g->enterpoint * g->d, g->d);
    (*dist_calc)++;
    pq_push(top, dist, g->enterpoint);
    pq_push(&cand, -dist, g->enterpoint);
    mass[g->enterpoint] = cv;
    float lower_bound = dist;

    while (cand.n) {
        orc_pair cur = pq_top(&cand);
        if (-cur.key > lower_bound) /* :67-68 */
            break;
        pq_pop(&cand);
        uint32_t node = cur.id;
        size_t cnt = g->counts[node];
        const uint32_t *nb = g->links + (size_t)node * g->maxM;
        for (size_t j = 0; j < cnt; j++) {
            uint32_t t = nb[j];
            if (mass[t] == cv)
                continue;
            mass[t] = cv;
            float dt = orc_l2sqr(point, g->vectors + (size_t)t * g->d, g->d);
            (*dist_calc)++;
            if (pq_top(top).key > dt || top->n < ef) { /* :93 */
                pq_push(&cand, -dt, t);
                pq_push(top, dt, t);
                if (top->n > ef)
                    pq_pop(top);
                lower_bound = pq_top(top).key;
            }
        }
    }
    pq_free(&cand);
}

/* hnswalg.cpp:112-146 getNeighborsByHeuristic. */
static void hnsw_neighbors_by_heuristic(orc_hnsw *g, orc_pq *top, size_t NN)
{
    if (top->n < NN)
        return;
    orc_pq closest;
    pq_init(&closest);
    while (top->n) {
        orc_pair p = pq_top(top);
        pq_push(&closest, -p.key, p.id);
        pq_pop(top);
    }
    orc_pair *keep = (orc_pair *)malloc(sizeof(orc_pair) * (NN ? NN : 1));
    size_t nkeep = 0;
    while (closest.n) {
        if (nkeep >= NN)
            break;
        orc_pair cur = pq_top(&closest);
        float dist_to_query = -cur.key;
        pq_pop(&closest);
        int good = 1;
        for (size_t i = 0; i < nkeep; i++) {
            float dd = orc_l2sqr(g->vectors + (size_t)keep[i].id * g->d,
                                 g->vectors + (size_t)cur.id * g->d, g->d);
            if (dd < dist_to_query) {
                good = 0;
                break;
            }
        }
        if (good)
            keep[nkeep++] = cur;
    }
    for (size_t i = 0; i < nkeep; i++)
        pq_push(top, -keep[i].key, keep[i].id);
    free(keep);
    pq_free(&closest);
}

/* hnswalg.cpp:148-210 mutuallyConnectNewElement. */
static int hnsw_connect(orc_hnsw *g, uint32_t cur_c, orc_pq *top)
{
    hnsw_neighbors_by_heuristic(g, top, g->M);
    uint32_t res[256];
    size_t nres = 0;
    while (top->n) {
        res[nres++] = pq_top(top).id;
        pq_pop(top);
    }
    g->counts[cur_c] = (uint8_t)nres;
    for (size_t i = 0; i < nres; i++)
        g->links[(size_t)cur_c * g->maxM + i] = res[i];

    for (size_t i = 0; i < nres; i++) {
        uint32_t o = res[i];
        if (o == cur_c)
            return -1;
        uint32_t *data = g->links + (size_t)o * g->maxM;
        size_t cnt = g->counts[o];
        if (cnt < g->maxM) {
            data[cnt] = cur_c;
            g->counts[o] = (uint8_t)(cnt + 1);
        } else {
            orc_pq cand;
            pq_init(&cand);
            float dmax = orc_l2sqr(g->vectors + (size_t)cur_c * g->d, g->vectors + (size_t)o * g->d, g->d);
            pq_push(&cand, dmax, cur_c);
            for (size_t j = 0; j < cnt; j++)
                pq_push(&cand,
                        orc_l2sqr(g->vectors + (size_t)data[j] * g->d, g->vectors + (size_t)o * g->d, g->d),
                        data[j]);
            hnsw_neighbors_by_heuristic(g, &cand, g->maxM);
            size_t w = 0;
            while (cand.n) {
                data[w++] = pq_top(&cand).id;
                pq_pop(&cand);
            }
            g->counts[o] = (uint8_t)w;
            pq_free(&cand);
        }
    }
    return 0;
}

/* hnswalg.cpp:212-225 addPoint. */
int orc_hnsw_add_point(orc_hnsw *g, const float *point)
{
    if (g->n >= g->maxelements)
        return -1;
    uint32_t cur_c = (uint32_t)g->n++;
    g->counts[cur_c] = 0;
    memset(g->links + (size_t)cur_c * g->maxM, 0, g->maxM * sizeof(uint32_t));
    memcpy(g->vectors + (size_t)cur_c * g->d, point, g->d * sizeof(float));
    if (cur_c == 0)
        return 0;
    orc_pq top;
    pq_init(&top);
    hnsw_search_base_layer(g, point, g->efConstruction, &top, g->visited, &g->epoch, &g->dist_calc);
    int rc = hnsw_connect(g, cur_c, &top);
    pq_free(&top);
    return rc;
}

/* hnswalg.cpp:227-234 searchKnn + the unload loop of IndexIVF_HNSW.cpp:249-259. */
static size_t hnsw_search_knn_tl(const orc_hnsw *g, const float *query, size_t ef, size_t k,
                                 uint32_t *out_ids, float *out_dists, uint16_t *mass, uint16_t *epoch,
                                 unsigned long long *dist_calc)
{
    orc_pq top;
    pq_init(&top);
    hnsw_search_base_layer(g, query, ef, &top, mass, epoch, dist_calc);
    while (top.n > k)
        pq_pop(&top);
    size_t r = top.n;
    for (size_t i = r; i-- > 0;) {
        out_dists[i] = pq_top(&top).key;
        out_ids[i] = pq_top(&top).id;
        pq_pop(&top);
    }
    pq_free(&top);
    return r;
}

size_t orc_hnsw_search_knn(orc_hnsw *g, const float *query, size_t ef, size_t k, uint32_t *out_ids,
                           float *out_dists)
{
    return hnsw_search_knn_tl(g, query, ef, k, out_ids, out_dists, g->visited, &g->epoch, &g->dist_calc);
}

/* hnswalg.cpp:236-265 SaveInfo / SaveEdges. */
int orc_hnsw_save(const orc_hnsw *g, const char *path_info, const char *path_edges)
{
    FILE *f = fopen(path_info, "wb");
    if (!f)
        return -1;
    size_t maxelements = g->n;
    size_t data_size = g->d * sizeof(float);
    size_t size_links_level0 = g->maxM * sizeof(uint32_t) + sizeof(uint8_t); /* hnswalg.cpp:25 */
    size_t offset_data = size_links_level0;
    size_t size_data_per_element = size_links_level0 + data_size;
    fwrite(&maxelements, sizeof(size_t), 1, f);
    fwrite(&g->enterpoint, sizeof(uint32_t), 1, f);
    fwrite(&data_size, sizeof(size_t), 1, f);
    fwrite(&offset_data, sizeof(size_t), 1, f);
    fwrite(&size_data_per_element, sizeof(size_t), 1, f);
    fwrite(&g->M, sizeof(size_t), 1, f);
    fwrite(&g->maxM, sizeof(size_t), 1, f);
    fwrite(&size_links_level0, sizeof(size_t), 1, f);
    fclose(f);
    f = fopen(path_edges, "wb");
    if (!f)
        return -1;
    for (size_t i = 0; i < g->n; i++) {
        uint32_t cnt = g->counts[i];
        fwrite(&cnt, sizeof(uint32_t), 1, f);
        fwrite(g->links + i * g->maxM, sizeof(uint32_t), cnt, f);
    }
    fclose(f);
    return 0;
}

/* hnswalg.cpp:267-324 LoadInfo / LoadData / LoadEdges. */
orc_hnsw *orc_hnsw_load(const char *path_info, const char *path_data, const char *path_edges)
{
    FILE *f = fopen(path_info, "rb");
    if (!f)
        return NULL;
    size_t maxelements, data_size, offset_data, size_data_per_element, M, maxM, size_links_level0;
    uint32_t enterpoint;
    int ok = 1;
    ok &= fread(&maxelements, sizeof(size_t), 1, f) == 1;
    ok &= fread(&enterpoint, sizeof(uint32_t), 1, f) == 1;
    ok &= fread(&data_size, sizeof(size_t), 1, f) == 1;
    ok &= fread(&offset_data, sizeof(size_t), 1, f) == 1;
    ok &= fread(&size_data_per_element, sizeof(size_t), 1, f) == 1;
    ok &= fread(&M, sizeof(size_t), 1, f) == 1;
    ok &= fread(&maxM, sizeof(size_t), 1, f) == 1;
    ok &= fread(&size_links_level0, sizeof(size_t), 1, f) == 1;
    fclose(f);
    if (!ok)
        return NULL;
    size_t d = data_size / sizeof(float);
    orc_hnsw *g = orc_hnsw_new(d, maxelements, M, maxM, 0);
    g->n = maxelements;
    g->enterpoint = enterpoint;
    f = fopen(path_data, "rb");
    if (!f) {
        orc_hnsw_free(g);
        return NULL;
    }
    for (size_t i = 0; i < maxelements; i++) {
        uint32_t dim;
        if (fread(&dim, sizeof(uint32_t), 1, f) != 1 || dim != d ||
            fread(g->vectors + i * d, sizeof(float), d, f) != d) {
            fclose(f);
            orc_hnsw_free(g);
            return NULL;
        }
    }
    fclose(f);
    f = fopen(path_edges, "rb");
    if (!f) {
        orc_hnsw_free(g);
        return NULL;
    }
    for (size_t i = 0; i < maxelements; i++) {
        uint32_t cnt;
        if (fread(&cnt, sizeof(uint32_t), 1, f) != 1 || cnt > maxM ||
            fread(g->links + i * maxM, sizeof(uint32_t), cnt, f) != cnt) {
            fclose(f);
            orc_hnsw_free(g);
            return NULL;
        }
        g->counts[i] = (uint8_t)cnt;
    }
    fclose(f);
    return g;
}

/* =============================================================================================
 * search
 * ============================================================================================= */

/* IndexIVF_HNSW.cpp:802-814 pq_L2sqr: sequential m = 0..code_size-1 (unrolled by 4 there). */
#ifndef ORC_OFAST_ORDER
static float adc_sum(const float *tab, const uint8_t *code, size_t code_size)
{
    float result = 0.0f;
    for (size_t m = 0; m < code_size; m++)
        result = result + tab[256 * m + code[m]];
    return result;
}
#else
/* g++ 11.4 -Ofast on that loop: result += (t0 + t1) + (t2 + t3) per group of four (see orc_l2sqr above). */
static float adc_sum(const float *tab, const uint8_t *code, size_t code_size)
{
    float result = 0.0f;
    for (size_t m = 0; m + 4 <= code_size; m += 4) {
        float a = tab[256 * m + code[m]] + tab[256 * (m + 1) + code[m + 1]];
        float b = tab[256 * (m + 2) + code[m + 2]] + tab[256 * (m + 3) + code[m + 3]];
        float c = a + b;
        result = result + c;
    }
    return result;
}
#endif

typedef struct {
    float *query;   /* rotated query (do_opq) */
    float *tab;     /* precomputed_table, IndexIVF_HNSW.h:183 */
    uint32_t *cidx; /* centroid_idxs */
    float *cdist;
    float *qcd;     /* Grouping.h:58 query_centroid_dists, nc floats, zero between queries */
    uint32_t *used; /* used_centroid_idxs */
    size_t nused, used_cap;
    float *qsd;     /* query_subcentroid_dists */
    uint16_t *visited;
    uint16_t epoch;
} orc_scratch;

static void scratch_init(orc_scratch *s, const orc_index *ix)
{
    memset(s, 0, sizeof(*s));
    s->query = (float *)malloc(ix->d * sizeof(float));
    s->tab = (float *)malloc(256 * ix->code_size * sizeof(float));
    s->cidx = (uint32_t *)malloc((ix->nprobe + 1) * sizeof(uint32_t));
    s->cdist = (float *)malloc((ix->nprobe + 1) * sizeof(float));
    if (ix->nsubc) {
        s->qcd = (float *)calloc(ix->nc, sizeof(float));
        s->used_cap = ix->nsubc * ix->nprobe * 2 + ix->nprobe + 16;
        s->used = (uint32_t *)malloc(s->used_cap * sizeof(uint32_t));
        s->qsd = (float *)malloc((ix->nsubc * ix->nprobe + 1) * sizeof(float));
    }
    if (ix->quantizer) {
        s->visited = (uint16_t *)calloc(ix->quantizer->maxelements, sizeof(uint16_t));
        s->epoch = (uint16_t)-1;
    }
}

static void scratch_free(orc_scratch *s)
{
    free(s->query);
    free(s->tab);
    free(s->cidx);
    free(s->cdist);
    free(s->qcd);
    free(s->used);
    free(s->qsd);
    free(s->visited);
}

static void used_push(orc_scratch *s, uint32_t c)
{
    if (s->nused == s->used_cap) {
        s->used_cap *= 2;
        s->used = (uint32_t *)realloc(s->used, s->used_cap * sizeof(uint32_t));
    }
    s->used[s->nused++] = c;
}

/* IndexIVF_HNSW.cpp:262-293: table, heapify, scan loop with the max_codes rule. */
static void ivf_scan(const orc_index *ix, size_t k, const orc_scratch *s, size_t nfound, float *distances,
                     long *labels, orc_stats *st)
{
    orc_inner_prod_table(s->query, ix->pq_centroids, ix->d, ix->code_size, s->tab);
    orc_maxheap_heapify(k, distances, labels);
    size_t ncode = 0;
    for (size_t i = 0; i < nfound; i++) {
        uint32_t c = s->cidx[i];
        size_t group_size = (size_t)(ix->offsets[c + 1] - ix->offsets[c]);
        if (group_size == 0)
            continue;
        const uint8_t *code = ix->codes + ix->offsets[c] * ix->code_size;
        const uint8_t *norm_code = ix->norm_codes + ix->offsets[c];
        const uint32_t *id = ix->ids + ix->offsets[c];
        float term1 = s->cdist[i] - ix->centroid_norms[c];
        for (size_t j = 0; j < group_size; j++) {
            float norm = ix->norm_table[norm_code[j]]; /* norm_pq->decode, :280 */
            float term3 = 2 * adc_sum(s->tab, code + j * ix->code_size, ix->code_size);
            float t = term1 + norm;
            float dist = t - term3; /* :284 */
            if (dist < distances[0]) {
                orc_maxheap_pop(k, distances, labels);
                orc_maxheap_push(k, distances, labels, dist, (long)id[j]);
            }
        }
        ncode += group_size;
        if (st)
            st->nseg++;
        if (ncode >= ix->max_codes)
            break;
    }
    if (st)
        st->ncode += ncode;
}

static const float *prepare_query(const orc_index *ix, const float *x, orc_scratch *s)
{
    if (ix->do_opq)
        orc_opq_apply(ix->opq_A, x, ix->d, s->query); /* :240 */
    else
        memcpy(s->query, x, ix->d * sizeof(float));
    return s->query;
}

static size_t coarse_stage(const orc_index *ix, orc_scratch *s, orc_stats *st)
{
    unsigned long long dc = 0;
    /* Precondition of the reference: efSearch >= nprobe and >= nprobe reachable nodes
     * (IndexIVF_HNSW.cpp:249-258 pops an empty queue otherwise).  Defined here: use what was found. */
    size_t r = hnsw_search_knn_tl(ix->quantizer, s->query, ix->efSearch, ix->nprobe, s->cidx, s->cdist,
                                  s->visited, &s->epoch, &dc);
    if (st)
        st->dist_evals += dc;
    return r;
}

static void search_ivf_s(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                         orc_stats *st, orc_scratch *s, const uint32_t *cidx, const float *cdist)
{
    prepare_query(ix, x, s);
    size_t nfound;
    if (cidx) {
        nfound = ix->nprobe;
        memcpy(s->cidx, cidx, nfound * sizeof(uint32_t));
        memcpy(s->cdist, cdist, nfound * sizeof(float));
    } else {
        nfound = coarse_stage(ix, s, st);
    }
    ivf_scan(ix, k, s, nfound, distances, labels, st);
}

/* Grouping.cpp:244-250 / :311-316: lazily evaluated distance to a neighbour centroid. */
static float group_qcd(const orc_index *ix, orc_scratch *s, uint32_t nn, orc_stats *st)
{
    if ((double)s->qcd[nn] < ORC_EPS) {
        s->qcd[nn] = orc_l2sqr(s->query, ix->quantizer->vectors + (size_t)nn * ix->d, ix->d);
        used_push(s, nn);
        if (st)
            st->dist_evals++;
    }
    return s->qcd[nn];
}

/* IndexIVF_HNSW_Grouping.cpp:188-363 after the coarse stage has filled s->cidx / s->cdist. */
static void grouping_scan(const orc_index *ix, size_t k, orc_scratch *s, size_t nfound, float *distances,
                          long *labels, orc_stats *st)
{
    const size_t nsubc = ix->nsubc;
    s->nused = 0;
    for (size_t i = nfound; i-- > 0;) { /* :207-220, farthest first */
        s->qcd[s->cidx[i]] = s->cdist[i];
        used_push(s, s->cidx[i]);
    }

    float threshold = 0.0f;
    if (ix->do_pruning) { /* :223-262 */
        size_t ncode = 0, nsubgroups = 0;
        for (size_t i = 0; i < nsubc * ix->nprobe; i++)
            s->qsd[i] = 0.0f; /* vector::resize value-initialises, :228 */
        float *qsd = s->qsd;
        for (size_t i = 0; i < nfound; i++) {
            uint32_t c = s->cidx[i];
            size_t group_size = (size_t)(ix->offsets[c + 1] - ix->offsets[c]);
            if (group_size == 0)
                continue;
            float alpha = ix->alphas[c];
            float term1 = (1 - alpha) * s->qcd[c];
            for (size_t subc = 0; subc < nsubc; subc++) {
                if (ix->subgroup_sizes[c * nsubc + subc] == 0)
                    continue;
                uint32_t nn = ix->nn_centroid_idxs[c * nsubc + subc];
                float qn = group_qcd(ix, s, nn, st);
                float a = (1 - alpha) * ix->inter_centroid_dists[c * nsubc + subc];
                float b = a - qn;
                float e = alpha * b;
                qsd[subc] = term1 - e; /* :251-252 */
                threshold = threshold + qsd[subc];
                nsubgroups++;
            }
            ncode += group_size;
            qsd += nsubc;
            if (ncode >= 2 * ix->max_codes)
                break;
        }
        threshold = threshold / (float)nsubgroups; /* :261 float /= size_t */
    }

    orc_inner_prod_table(s->query, ix->pq_centroids, ix->d, ix->code_size, s->tab); /* :265 */
    orc_maxheap_heapify(k, distances, labels);

    size_t ncode = 0;
    const float *qsd = s->qsd;
    for (size_t i = 0; i < nfound; i++) { /* :283-353 */
        uint32_t c = s->cidx[i];
        size_t group_size = (size_t)(ix->offsets[c + 1] - ix->offsets[c]);
        if (group_size == 0)
            continue;
        float alpha = ix->alphas[c];
        float term1 = (1 - alpha) * (s->qcd[c] - ix->centroid_norms[c]);
        const uint8_t *code = ix->codes + ix->offsets[c] * ix->code_size;
        const uint8_t *norm_code = ix->norm_codes + ix->offsets[c];
        const uint32_t *id = ix->ids + ix->offsets[c];
        for (size_t subc = 0; subc < nsubc; subc++) {
            size_t sg = ix->subgroup_sizes[c * nsubc + subc];
            if (sg == 0)
                continue;
            if (!ix->do_pruning || qsd[subc] < threshold) { /* :308 */
                uint32_t nn = ix->nn_centroid_idxs[c * nsubc + subc];
                float qn = group_qcd(ix, s, nn, st);
                float term2 = alpha * (qn - ix->centroid_norms[nn]);
                for (size_t j = 0; j < sg; j++) {
                    float norm = ix->norm_table[norm_code[j]];
                    float term4 = 2 * adc_sum(s->tab, code + j * ix->code_size, ix->code_size);
                    float t = term1 + term2;
                    t = t + norm;
                    float dist = t - term4; /* :323 */
                    if (dist < distances[0]) {
                        orc_maxheap_pop(k, distances, labels);
                        orc_maxheap_push(k, distances, labels, dist, (long)id[j]);
                    }
                }
                ncode += sg;
                if (st)
                    st->nseg++;
            }
            code += sg * ix->code_size;
            norm_code += sg;
            id += sg;
        }
        if (ncode >= ix->max_codes)
            break;
        if (ix->do_pruning)
            qsd += nsubc;
    }
    if (st)
        st->ncode += ncode;
    for (size_t i = 0; i < s->nused; i++) /* :358-359 */
        s->qcd[s->used[i]] = 0;
}

static void search_grouping_s(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                              orc_stats *st, orc_scratch *s, const uint32_t *cidx, const float *cdist)
{
    prepare_query(ix, x, s);
    size_t nfound;
    if (cidx) {
        nfound = ix->nprobe;
        memcpy(s->cidx, cidx, nfound * sizeof(uint32_t));
        memcpy(s->cdist, cdist, nfound * sizeof(float));
    } else {
        nfound = coarse_stage(ix, s, st);
    }
    grouping_scan(ix, k, s, nfound, distances, labels, st);
}

void orc_search_ivf(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                    orc_stats *st)
{
    orc_scratch s;
    scratch_init(&s, ix);
    search_ivf_s(ix, k, x, distances, labels, st, &s, NULL, NULL);
    scratch_free(&s);
}

void orc_search_ivf_coarse(const orc_index *ix, size_t k, const float *x, const uint32_t *centroid_idxs,
                           const float *query_centroid_dists, float *distances, long *labels, orc_stats *st)
{
    orc_scratch s;
    scratch_init(&s, ix);
    search_ivf_s(ix, k, x, distances, labels, st, &s, centroid_idxs, query_centroid_dists);
    scratch_free(&s);
}

void orc_search_grouping(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                         orc_stats *st)
{
    orc_scratch s;
    scratch_init(&s, ix);
    search_grouping_s(ix, k, x, distances, labels, st, &s, NULL, NULL);
    scratch_free(&s);
}

void orc_search_grouping_coarse(const orc_index *ix, size_t k, const float *x, const uint32_t *centroid_idxs,
                                const float *coarse_dists, float *distances, long *labels, orc_stats *st)
{
    orc_scratch s;
    scratch_init(&s, ix);
    search_grouping_s(ix, k, x, distances, labels, st, &s, centroid_idxs, coarse_dists);
    scratch_free(&s);
}

void orc_search_batch(const orc_index *ix, size_t nq, size_t k, const float *x, float *distances,
                      long *labels, uint32_t *out_coarse_ids, float *out_coarse_dists, orc_stats *st_sum,
                      int nthreads)
{
    orc_stats total = {0, 0, 0};
    if (nthreads < 1)
        nthreads = 1;
#pragma omp parallel num_threads(nthreads) if (nthreads > 1)
    {
        orc_scratch s;
        scratch_init(&s, ix);
        orc_stats st = {0, 0, 0};
#pragma omp for schedule(dynamic, 16)
        for (long long qi = 0; qi < (long long)nq; qi++) {
            const float *xq = x + (size_t)qi * ix->d;
            prepare_query(ix, xq, &s);
            size_t nfound = coarse_stage(ix, &s, &st);
            if (out_coarse_ids)
                for (size_t i = 0; i < ix->nprobe; i++) {
                    out_coarse_ids[(size_t)qi * ix->nprobe + i] = i < nfound ? s.cidx[i] : 0xffffffffu;
                    out_coarse_dists[(size_t)qi * ix->nprobe + i] = i < nfound ? s.cdist[i] : 0.0f;
                }
            if (ix->nsubc)
                grouping_scan(ix, k, &s, nfound, distances + (size_t)qi * k, labels + (size_t)qi * k, &st);
            else
                ivf_scan(ix, k, &s, nfound, distances + (size_t)qi * k, labels + (size_t)qi * k, &st);
        }
#pragma omp critical
        {
            total.ncode += st.ncode;
            total.nseg += st.nseg;
            total.dist_evals += st.dist_evals;
        }
        scratch_free(&s);
    }
    if (st_sum)
        *st_sum = total;
}

/* IndexIVF_HNSW.cpp:781-787; faiss spec fvec_norm_L2sqr = inner product of x with itself (SSE order). */
void orc_compute_centroid_norms(const orc_hnsw *g, float *centroid_norms)
{
    for (size_t i = 0; i < g->n; i++)
        centroid_norms[i] = orc_inner_product_sse_order(g->vectors + i * g->d, g->vectors + i * g->d, g->d);
}

/* IndexIVF_HNSW_Grouping.cpp:620-631. */
void orc_compute_inter_centroid_dists(const orc_hnsw *g, size_t nsubc, const uint32_t *nn_idx, float *out)
{
    for (size_t i = 0; i < g->n; i++)
        for (size_t subc = 0; subc < nsubc; subc++)
            out[i * nsubc + subc] = orc_l2sqr(g->vectors + (size_t)nn_idx[i * nsubc + subc] * g->d,
                                              g->vectors + i * g->d, g->d);
}

/* IndexIVF_HNSW.cpp:789-800. */
void orc_rotate_quantizer(orc_hnsw *g, const float *A)
{
    float *tmp = (float *)malloc(g->d * sizeof(float));
    for (size_t i = 0; i < g->n; i++) {
        memcpy(tmp, g->vectors + i * g->d, g->d * sizeof(float));
        orc_opq_apply(A, tmp, g->d, g->vectors + i * g->d);
    }
    free(tmp);
}

/* =============================================================================================
 * Construction side: IndexIVF_HNSW::add_batch, IndexIVF_HNSW.cpp:75-121.
 * faiss leafs (spec of its SSE build, absent from the reference tree -- parity unpinned):
 *   fvec_madd(n, a, bf, b, c): c = a + bf * b, mul then add;
 *   fvec_L2sqr: 4 partial sums over blocks of 4, zero-padded tail, (s0+s1)+(s2+s3);
 *   ProductQuantizer::compute_code: per sub-quantizer the first c with dis < mindis, mindis starting
 *   at 1e20 and the index at -1 (stored as uint8);  decode: concatenated code words;
 *   LinearTransform::apply / transform_transpose: sgemm in faiss, order unspecified -- the k-ordered
 *   fmaf chain is the contract here (orc_opq_apply);  fvec_norm_L2sqr = inner product order.
 * ============================================================================================= */
static float l2_sse_order(const float *x, const float *y, size_t d)
{
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int l = 0; l < 4; l++) {
            float t = x[i + l] - y[i + l];
            s[l] = s[l] + t * t;
        }
    for (int l = 0; i + l < d; l++) {
        float t = x[i + l] - y[i + l];
        s[l] = s[l] + t * t;
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

static void pq_compute_code(const float *cb, size_t M, size_t dsub, const float *x, uint8_t *code)
{
    for (size_t m = 0; m < M; m++) {
        float mindis = 1e20f;
        int idxm = -1;
        for (size_t c = 0; c < 256; c++) {
            float dis = l2_sse_order(x + m * dsub, cb + (m * 256 + c) * dsub, dsub);
            if (dis < mindis) {
                mindis = dis;
                idxm = (int)c;
            }
        }
        code[m] = (uint8_t)idxm;
    }
}

void orc_add_batch_encode(const orc_index *ix, size_t n, const float *x, const uint32_t *precomputed_idx,
                          uint32_t *out_idx, uint8_t *out_codes, uint8_t *out_norm_codes, float *out_norms)
{
    const size_t d = ix->d, M = ix->code_size, dsub = d / M;
    const orc_hnsw *g = ix->quantizer;
#pragma omp parallel
    {
        float *res = (float *)malloc(3 * d * sizeof(float)), *tmp = res + d, *rec = res + 2 * d;
        /* every concurrent caller of searchKnn gets its own visited list (visited_list_pool.h:55-76) */
        uint16_t *mass = precomputed_idx ? NULL : (uint16_t *)calloc(g->maxelements, sizeof(uint16_t));
        uint16_t epoch = (uint16_t)-1;
        unsigned long long dc = 0;
#pragma omp for schedule(static)
        for (long ii = 0; ii < (long)n; ii++) {
            const size_t i = (size_t)ii;
            const float *xi = x + i * d;
            uint32_t key = 0;
            if (precomputed_idx) {
                key = precomputed_idx[i]; /* :79-80 */
            } else {
                float dist;
                hnsw_search_knn_tl(g, xi, ix->efSearch, 1, &key, &dist, mass, &epoch, &dc); /* :68-72 */
            }
            if (out_idx)
                out_idx[i] = key;
            const float *cen = g->vectors + (size_t)key * d;
            for (size_t j = 0; j < d; j++) /* :86-87 compute_residuals -> fvec_madd(d, x, -1, centroid, r) */
                res[j] = xi[j] + -1.0f * cen[j];
            const float *enc = res;
            if (ix->do_opq) { /* :90-94 */
                orc_opq_apply(ix->opq_A, res, d, tmp);
                enc = tmp;
            }
            uint8_t *code = out_codes + i * M;
            pq_compute_code(ix->pq_centroids, M, dsub, enc, code); /* :97-98 */
            float *dec = ix->do_opq ? res : tmp;
            for (size_t m = 0; m < M; m++) /* :101-102 pq->decode */
                memcpy(dec + m * dsub, ix->pq_centroids + (m * 256 + code[m]) * dsub, dsub * sizeof(float));
            const float *back = dec;
            if (ix->do_opq) { /* :105-109 transform_transpose: x[k] = sum_i A[i][k] * y[i] */
                for (size_t k = 0; k < d; k++) {
                    float acc = 0.0f;
                    for (size_t r = 0; r < d; r++)
                        acc = fmaf(ix->opq_A[r * d + k], dec[r], acc);
                    tmp[k] = acc;
                }
                back = tmp;
            }
            for (size_t j = 0; j < d; j++) /* :112-113 reconstruct -> fvec_madd(d, decoded, 1, centroid, x) */
                rec[j] = back[j] + 1.0f * cen[j];
            float norm = orc_inner_product_sse_order(rec, rec, d); /* :116-117 */
            if (out_norms)
                out_norms[i] = norm;
            pq_compute_code(ix->norm_table, 1, 1, &norm, out_norm_codes + i); /* :120-121 */
        }
        free(res);
        free(mass);
    }
}

/* =============================================================================================
 * Code book training, IndexIVF_HNSW.cpp:536-593 (train_pq): the reference hands residuals to
 * faiss::ProductQuantizer::train, i.e. faiss's Clustering per sub-space.  faiss is absent from the
 * reference tree (parity unpinned): restated here is the Lloyd iteration this repo's host classes run
 * (ivf-hnsw_amd/csrc/host/faiss_min.cpp) and the device path must reproduce bit for bit --
 *   assignment: the first nearest code word in fvec_L2sqr's SSE order, as compute_code finds it;
 *   update    : mean of the assigned sub-vectors, the sum taken in point order in float, divided by
 *               the count; a code word nothing was assigned to stays as it is.
 * ============================================================================================= */
void orc_pq_lloyd(size_t n, size_t d, size_t M, const float *x, size_t niter, float *centroids,
                  uint8_t *out_assign)
{
    const size_t dsub = d / M;
    uint8_t *assign = (uint8_t *)malloc(n * M + 1);
    float *sum = (float *)malloc(256 * dsub * sizeof(float));
    size_t *cnt = (size_t *)malloc(256 * sizeof(size_t));
    for (size_t it = 0; it < niter; it++) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; i++)
            pq_compute_code(centroids, M, dsub, x + (size_t)i * d, assign + (size_t)i * M);
        for (size_t m = 0; m < M; m++) {
            memset(sum, 0, 256 * dsub * sizeof(float));
            memset(cnt, 0, 256 * sizeof(size_t));
            for (size_t i = 0; i < n; i++) {
                const size_t c = assign[i * M + m];
                cnt[c]++;
                for (size_t j = 0; j < dsub; j++)
                    sum[c * dsub + j] = sum[c * dsub + j] + x[i * d + m * dsub + j];
            }
            for (size_t c = 0; c < 256; c++)
                if (cnt[c])
                    for (size_t j = 0; j < dsub; j++)
                        centroids[(m * 256 + c) * dsub + j] = sum[c * dsub + j] / (float)cnt[c];
        }
    }
    if (out_assign)
        memcpy(out_assign, assign, n * M);
    free(assign);
    free(sum);
    free(cnt);
}

/* The d x d product of OPQ's Procrustes step, C[a][b] = sum_i X[i][a] * Y[i][b] (faiss OPQMatrix::train hands it
 * to sgemm: order unspecified).  The contract here is the order the device's MFMA kernel produces: points in chunks
 * of `chunk`, inside a chunk a fmaf chain in point order starting from 0 (v_mfma_f32_32x32x2_f32 is exactly that),
 * the chunks' partial products added in chunk order. */
void orc_xty(size_t n, size_t d, const float *X, const float *Y, size_t chunk, float *C)
{
#pragma omp parallel for schedule(static)
    for (long aa = 0; aa < (long)d; aa++) {
        const size_t a = (size_t)aa;
        for (size_t b = 0; b < d; b++) {
            float total = 0.0f;
            for (size_t i0 = 0; i0 < n; i0 += chunk) {
                const size_t i1 = i0 + chunk < n ? i0 + chunk : n;
                float acc = 0.0f;
                for (size_t i = i0; i < i1; i++)
                    acc = fmaf(X[i * d + a], Y[i * d + b], acc);
                total = i0 == 0 ? acc : total + acc;
            }
            C[a * d + b] = total;
        }
    }
}

/* IndexIVF_HNSW_Grouping.cpp:691-733 compute_alpha.  The per-point winner is maxheap.top() of
 * pair<-dist, pair<numerator, denominator>>: smallest dist, ties to the larger numerator, then denominator. */
static float grouping_compute_alpha(size_t d, size_t nsubc, const float *centroid_vectors, const float *points,
                                    const float *centroid, const float *cv_norms, size_t group_size)
{
    float group_numerator = 0.0f, group_denominator = 0.0f;
    float *pv = (float *)malloc(2 * d * sizeof(float)), *sub = pv + d;
    for (size_t i = 0; i < group_size; i++) {
        const float *point = points + i * d;
        for (size_t j = 0; j < d; j++) /* :699-700 fvec_madd(d, point, -1, centroid, point_vector) */
            pv[j] = point[j] + -1.0f * centroid[j];
        int have = 0;
        float bneg = 0.0f, bnum = 0.0f, bden = 0.0f;
        for (size_t s = 0; s < nsubc; s++) {
            const float *cv = centroid_vectors + s * d;
            float numerator = orc_inner_product_sse_order(cv, pv, d); /* :711 */
            numerator = (numerator > 0) ? numerator : 0.0f;          /* :712 */
            const float denominator = cv_norms[s];
            const float a = numerator / denominator;
            for (size_t j = 0; j < d; j++) /* :718 fvec_madd(d, centroid, alpha, cv, subcentroid) */
                sub[j] = centroid[j] + a * cv[j];
            const float neg = -orc_l2sqr(point, sub, d); /* :720-721 */
            int better;
            if (!have)
                better = 1;
            else if (neg != neg)
                better = 0; /* NaN never displaces anything */
            else if (bneg != bneg)
                better = 1;
            else /* std::pair operator< on (neg, (num, den)) */
                better = bneg < neg ||
                         (!(neg < bneg) && (bnum < numerator || (!(numerator < bnum) && bden < denominator)));
            if (better) {
                have = 1;
                bneg = neg;
                bnum = numerator;
                bden = denominator;
            }
        }
        group_numerator += bnum; /* :724-725 */
        group_denominator += bden;
    }
    free(pv);
    return (group_denominator > 0) ? group_numerator / group_denominator : 0.0f; /* :727 */
}

int orc_add_group_encode(const orc_index *ix, size_t nsubc, uint32_t centroid_idx, size_t group_size,
                         const float *data, uint32_t *nn_centroid_idxs, float *alpha, uint32_t *subcentroid_idxs,
                         uint8_t *codes, uint8_t *norm_codes)
{
    const size_t d = ix->d, M = ix->code_size, dsub = d / M;
    orc_hnsw *g = ix->quantizer;
    const float *centroid = g->vectors + (size_t)centroid_idx * d;
    /* :47-62 nearest first; the nearest one (the centroid itself) is dropped */
    uint32_t *ids = (uint32_t *)malloc((nsubc + 1) * sizeof(uint32_t));
    float *dists = (float *)malloc((nsubc + 1) * sizeof(float));
    size_t r = orc_hnsw_search_knn(g, centroid, ix->efSearch, nsubc + 1, ids, dists);
    if (r != nsubc + 1) {
        free(ids);
        free(dists);
        return -1;
    }
    float *cv_norms = (float *)malloc(nsubc * sizeof(float));
    for (size_t s = 0; s < nsubc; s++) {
        nn_centroid_idxs[s] = ids[s + 1];
        cv_norms[s] = dists[s + 1];
    }
    free(ids);
    free(dists);
    if (group_size == 0) { /* :63-64 */
        free(cv_norms);
        return 0;
    }
    float *cvs = (float *)malloc(2 * nsubc * d * sizeof(float)), *subc = cvs + nsubc * d;
    for (size_t s = 0; s < nsubc; s++) { /* :70-74 fvec_madd(d, neighbour, -1, centroid, cv) */
        const float *nb = g->vectors + (size_t)nn_centroid_idxs[s] * d;
        for (size_t j = 0; j < d; j++)
            cvs[s * d + j] = nb[j] + -1.0f * centroid[j];
    }
    *alpha = grouping_compute_alpha(d, nsubc, cvs, data, centroid, cv_norms, group_size); /* :77-79 */
    for (size_t s = 0; s < nsubc; s++) /* :82-87 fvec_madd(d, centroid, alpha, cv, subcentroid) */
        for (size_t j = 0; j < d; j++)
            subc[s * d + j] = centroid[j] + *alpha * cvs[s * d + j];
    float *res = (float *)malloc(3 * d * sizeof(float)), *tmp = res + d, *rec = res + 2 * d;
    for (size_t i = 0; i < group_size; i++) {
        const float *xi = data + i * d;
        /* :673-689 first minimum of fvec_L2sqr(subcentroid, x) */
        float min_dist = 0.0f;
        long min_idx = -1;
        for (size_t s = 0; s < nsubc; s++) {
            float dist = orc_l2sqr(subc + s * d, xi, d);
            if (min_idx == -1 || dist < min_dist) {
                min_dist = dist;
                min_idx = (long)s;
            }
        }
        subcentroid_idxs[i] = (uint32_t)min_idx;
        const float *sc = subc + (size_t)min_idx * d;
        for (size_t j = 0; j < d; j++) /* :655-662 */
            res[j] = xi[j] + -1.0f * sc[j];
        const float *enc = res;
        if (ix->do_opq) { /* :97-101 */
            orc_opq_apply(ix->opq_A, res, d, tmp);
            enc = tmp;
        }
        uint8_t *code = codes + i * M;
        pq_compute_code(ix->pq_centroids, M, dsub, enc, code); /* :104-105 */
        float *dec = ix->do_opq ? res : tmp;
        for (size_t m = 0; m < M; m++) /* :108-109 */
            memcpy(dec + m * dsub, ix->pq_centroids + (m * 256 + code[m]) * dsub, dsub * sizeof(float));
        const float *back = dec;
        if (ix->do_opq) { /* :112-116 */
            for (size_t k = 0; k < d; k++) {
                float acc = 0.0f;
                for (size_t q = 0; q < d; q++)
                    acc = fmaf(ix->opq_A[q * d + k], dec[q], acc);
                tmp[k] = acc;
            }
            back = tmp;
        }
        for (size_t j = 0; j < d; j++) /* :664-671 */
            rec[j] = back[j] + 1.0f * sc[j];
        float norm = orc_inner_product_sse_order(rec, rec, d); /* :123-124 */
        pq_compute_code(ix->norm_table, 1, 1, &norm, norm_codes + i); /* :127-128 */
    }
    free(res);
    free(cvs);
    free(cv_norms);
    return 0;
}

/* =============================================================================================
 * .index files -- utils.h:53-81 (uint32 count + raw elements), IndexIVF_HNSW.cpp:637-663,758-779,
 * IndexIVF_HNSW_Grouping.cpp:397-483
 * ============================================================================================= */
static int wvec(FILE *f, const void *p, uint32_t n, size_t elt)
{
    if (fwrite(&n, sizeof(uint32_t), 1, f) != 1)
        return -1;
    if (n && fwrite(p, elt, n, f) != n)
        return -1;
    return 0;
}

int orc_index_write(const orc_index *ix, const char *path, int grouping)
{
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    int rc = 0;
    rc |= fwrite(&ix->d, sizeof(size_t), 1, f) != 1;
    rc |= fwrite(&ix->nc, sizeof(size_t), 1, f) != 1;
    if (grouping)
        rc |= fwrite(&ix->nsubc, sizeof(size_t), 1, f) != 1;
    for (size_t c = 0; c < ix->nc; c++)
        rc |= wvec(f, ix->ids + ix->offsets[c], (uint32_t)(ix->offsets[c + 1] - ix->offsets[c]), 4);
    for (size_t c = 0; c < ix->nc; c++)
        rc |= wvec(f, ix->codes + ix->offsets[c] * ix->code_size,
                   (uint32_t)((ix->offsets[c + 1] - ix->offsets[c]) * ix->code_size), 1);
    for (size_t c = 0; c < ix->nc; c++)
        rc |= wvec(f, ix->norm_codes + ix->offsets[c], (uint32_t)(ix->offsets[c + 1] - ix->offsets[c]), 1);
    if (grouping) {
        for (size_t c = 0; c < ix->nc; c++)
            rc |= wvec(f, ix->nn_centroid_idxs + c * ix->nsubc, (uint32_t)ix->nsubc, 4);
        for (size_t c = 0; c < ix->nc; c++) {
            /* empty groups keep an empty subgroup_sizes vector (Grouping.cpp:64-65,148) */
            uint32_t n = ix->offsets[c + 1] == ix->offsets[c] ? 0 : (uint32_t)ix->nsubc;
            rc |= wvec(f, ix->subgroup_sizes + c * ix->nsubc, n, 4);
        }
        rc |= wvec(f, ix->alphas, (uint32_t)ix->nc, 4);
    }
    rc |= wvec(f, ix->centroid_norms, (uint32_t)ix->nc, 4);
    if (grouping)
        for (size_t c = 0; c < ix->nc; c++)
            rc |= wvec(f, ix->inter_centroid_dists + c * ix->nsubc, (uint32_t)ix->nsubc, 4);
    fclose(f);
    return rc ? -1 : 0;
}

static int rcount(FILE *f, uint32_t *n) { return fread(n, sizeof(uint32_t), 1, f) == 1 ? 0 : -1; }

int orc_index_read(orc_index *ix, const char *path, int grouping)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return -1;
    size_t d, nc, nsubc = 0;
    if (fread(&d, sizeof(size_t), 1, f) != 1 || fread(&nc, sizeof(size_t), 1, f) != 1)
        goto fail;
    if (grouping && fread(&nsubc, sizeof(size_t), 1, f) != 1)
        goto fail;
    ix->d = d;
    ix->nc = nc;
    ix->nsubc = nsubc;
    ix->offsets = (uint64_t *)calloc(nc + 1, sizeof(uint64_t));
    /* pass 1 over the id vectors to size the CSR arrays */
    long ids_pos = ftell(f);
    for (size_t c = 0; c < nc; c++) {
        uint32_t n;
        if (rcount(f, &n))
            goto fail;
        ix->offsets[c + 1] = ix->offsets[c] + n;
        if (fseek(f, (long)n * 4, SEEK_CUR))
            goto fail;
    }
    size_t N = (size_t)ix->offsets[nc];
    ix->ids = (uint32_t *)malloc((N ? N : 1) * 4);
    ix->norm_codes = (uint8_t *)malloc(N ? N : 1);
    fseek(f, ids_pos, SEEK_SET);
    for (size_t c = 0; c < nc; c++) {
        uint32_t n;
        if (rcount(f, &n) || fread(ix->ids + ix->offsets[c], 4, n, f) != n)
            goto fail;
    }
    ix->codes = NULL;
    for (size_t c = 0; c < nc; c++) {
        uint32_t n;
        if (rcount(f, &n))
            goto fail;
        size_t ln = (size_t)(ix->offsets[c + 1] - ix->offsets[c]);
        if (ln) {
            if (n % ln)
                goto fail;
            size_t cs = n / ln;
            if (!ix->codes) {
                ix->code_size = cs;
                ix->codes = (uint8_t *)malloc(N * cs);
            } else if (cs != ix->code_size)
                goto fail;
            if (fread(ix->codes + ix->offsets[c] * cs, 1, n, f) != n)
                goto fail;
        } else if (n)
            goto fail;
    }
    if (!ix->codes)
        ix->codes = (uint8_t *)malloc(1);
    for (size_t c = 0; c < nc; c++) {
        uint32_t n;
        if (rcount(f, &n) || n != ix->offsets[c + 1] - ix->offsets[c] ||
            fread(ix->norm_codes + ix->offsets[c], 1, n, f) != n)
            goto fail;
    }
    if (grouping) {
        ix->nn_centroid_idxs = (uint32_t *)calloc(nc * nsubc + 1, 4);
        ix->subgroup_sizes = (uint32_t *)calloc(nc * nsubc + 1, 4);
        ix->inter_centroid_dists = (float *)calloc(nc * nsubc + 1, 4);
        ix->alphas = (float *)calloc(nc + 1, 4);
        for (size_t c = 0; c < nc; c++) {
            uint32_t n;
            if (rcount(f, &n) || (n != 0 && n != nsubc) || fread(ix->nn_centroid_idxs + c * nsubc, 4, n, f) != n)
                goto fail;
        }
        for (size_t c = 0; c < nc; c++) {
            uint32_t n;
            if (rcount(f, &n) || (n != 0 && n != nsubc) || fread(ix->subgroup_sizes + c * nsubc, 4, n, f) != n)
                goto fail;
        }
        uint32_t n;
        if (rcount(f, &n) || n != nc || fread(ix->alphas, 4, n, f) != n)
            goto fail;
    }
    {
        uint32_t n;
        ix->centroid_norms = (float *)malloc((nc + 1) * 4);
        if (rcount(f, &n) || n != nc || fread(ix->centroid_norms, 4, n, f) != n)
            goto fail;
    }
    if (grouping)
        for (size_t c = 0; c < nc; c++) {
            uint32_t n;
            if (rcount(f, &n) || (n != 0 && n != nsubc) ||
                fread(ix->inter_centroid_dists + c * nsubc, 4, n, f) != n)
                goto fail;
        }
    fclose(f);
    return 0;
fail:
    fclose(f);
    return -1;
}

void orc_index_free_lists(orc_index *ix)
{
    free(ix->offsets);
    free(ix->ids);
    free(ix->codes);
    free(ix->norm_codes);
    free(ix->centroid_norms);
    free(ix->alphas);
    free(ix->nn_centroid_idxs);
    free(ix->subgroup_sizes);
    free(ix->inter_centroid_dists);
    ix->offsets = NULL;
    ix->ids = NULL;
    ix->codes = ix->norm_codes = NULL;
    ix->centroid_norms = ix->alphas = ix->inter_centroid_dists = NULL;
    ix->nn_centroid_idxs = ix->subgroup_sizes = NULL;
}

/* =============================================================================================
 * exact k-nearest-neighbour tables (the contract of ivfhnsw_gpu_knn, kernels_knn.hip): NOT a reference
 * function -- the exact form of what hnswalg.cpp:112-225 and IndexIVF_HNSW_Grouping.cpp:47-62 approximate
 * with graph searches, and of the drivers' ground-truth files.  norm and dot are fmaf chains over
 * k = 0..d-1 (the order v_mfma_f32_32x32x2_f32 accumulates in), dist = (norm(q) + norm(x)) - 2 * dot.
 * queries == NULL: the base rows themselves.  ids / dists: [nq][k] ascending by (dist, id); missing
 * slots 0xffffffff / FLT_MAX.
 * ============================================================================================= */
void orc_knn(size_t nq, size_t nx, size_t d, const float *queries, const float *base, size_t k, int mode,
             uint32_t *ids, float *dists)
{
    /* mode: 0 every base row is a candidate, 1 not row q itself, 2 only rows before q (ivfhnsw_hip.h IVFHNSW_KNN_*) */
    if (queries == NULL) {
        queries = base;
        nq = nx;
    }
    float *xn = (float *)malloc((nx ? nx : 1) * sizeof(float));
    for (size_t i = 0; i < nx; i++) {
        float acc = 0.f;
        for (size_t kk = 0; kk < d; kk++)
            acc = fmaf(base[i * d + kk], base[i * d + kk], acc);
        xn[i] = acc;
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (long q = 0; q < (long)nq; q++) {
        const float *qv = queries + (size_t)q * d;
        float qn = 0.f;
        for (size_t kk = 0; kk < d; kk++)
            qn = fmaf(qv[kk], qv[kk], qn);
        uint32_t *oi = ids + (size_t)q * k;
        float *od = dists + (size_t)q * k;
        size_t have = 0;
        for (size_t i = 0; i < nx; i++) {
            if ((mode == 1 && i == (size_t)q) || (mode == 2 && i >= (size_t)q))
                continue;
            float dot = 0.f;
            for (size_t kk = 0; kk < d; kk++)
                dot = fmaf(qv[kk], base[i * d + kk], dot);
            float t = qn + xn[i];
            float dist = t - 2.0f * dot;
            /* insertion into the ascending (dist, id) list; ids arrive ascending, so strict '<' keeps the earlier id */
            if (have == k && !(dist < od[k - 1]))
                continue;
            size_t pos = have < k ? have : k - 1;
            while (pos > 0 && dist < od[pos - 1]) {
                od[pos] = od[pos - 1];
                oi[pos] = oi[pos - 1];
                pos--;
            }
            od[pos] = dist;
            oi[pos] = (uint32_t)i;
            if (have < k)
                have++;
        }
        for (size_t j = have; j < k; j++) {
            oi[j] = 0xffffffffu;
            od[j] = FLT_MAX;
        }
    }
    free(xn);
}

/* The serial loop ivfhnsw_gpu_build_graph unrolls: hnswalg.cpp:212-225 addPoint for c = 0..n-1, where the
 * candidates handed to mutuallyConnectNewElement are the exact ncand nearest earlier nodes (orc_knn, mode 2)
 * with their fstdistfunc distances, instead of searchBaseLayer's results.  Everything else is the
 * reference's construction code above (hnsw_connect). */
orc_hnsw *orc_hnsw_build_exact(size_t d, size_t n, size_t M, size_t maxM, size_t ncand, const float *vectors)
{
    orc_hnsw *g = orc_hnsw_new(d, n, M, maxM, ncand);
    memcpy(g->vectors, vectors, n * d * sizeof(float));
    uint32_t *ids = (uint32_t *)malloc((n ? n : 1) * ncand * sizeof(uint32_t));
    float *dd = (float *)malloc((n ? n : 1) * ncand * sizeof(float));
    orc_knn(0, n, d, NULL, vectors, ncand, 2, ids, dd);
    for (size_t c = 0; c < n; c++) {
        g->n = c + 1;
        if (c == 0)
            continue;
        orc_pq top;
        pq_init(&top);
        for (size_t i = 0; i < ncand && ids[c * ncand + i] != 0xffffffffu; i++) {
            uint32_t t = ids[c * ncand + i];
            pq_push(&top, orc_l2sqr(vectors + c * d, vectors + (size_t)t * d, d), t);
        }
        hnsw_connect(g, (uint32_t)c, &top);
        pq_free(&top);
    }
    free(ids);
    free(dd);
    return g;
}
