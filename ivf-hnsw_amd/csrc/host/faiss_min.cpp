// Independent implementation of the small faiss surface the reference links against.  faiss
// itself is an un-vendored submodule of the reference and is not available; algorithms follow faiss's
// published behaviour.  The search hot path does NOT run through here -- it runs on the device -- these
// functions serve file I/O, centroid rotation at load time and construction-side callers.
//
// Built with -ffp-contract=off: the float orders below are the ones the device kernels use.
#include <faiss/Heap.h>
#include <faiss/ProductQuantizer.h>
#include <faiss/VectorTransform.h>
#include <faiss/index_io.h>
#include <faiss/utils.h>

#include <ivfhnsw_hip.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>

namespace faiss {

namespace {

// Training runs on the device (ivfhnsw_gpu_pq_train, ivfhnsw_gpu_xty) through ONE handle per process, created at the
// first training call and kept (its stream, status word and the t_x / t_y buffers are reused by every Lloyd batch and
// every OPQ iteration; round 2 created and destroyed a handle per iteration).  Never destroyed: a static destructor
// would run after the HIP runtime's own.  There is no host fallback: without a gfx950 device training fails, like every
// other device entry point.
struct TrainDevice {
    ivfhnsw_gpu *h = nullptr;
    TrainDevice()
    {
        static ivfhnsw_gpu *shared = [] {
            ivfhnsw_gpu *p = nullptr;
            if (ivfhnsw_gpu_create(0, &p))
                p = nullptr;
            return p;
        }();
        if (!shared)
            throw std::runtime_error(std::string("code-book training needs the device: ") + ivfhnsw_gpu_last_error());
        h = shared;
    }
    TrainDevice(const TrainDevice &) = delete;
    TrainDevice &operator=(const TrainDevice &) = delete;
};

} // namespace

// ---------------------------------------------------------------------------------------------- heap
void maxheap_heapify(size_t k, float *vals, long *ids, const float *x, const long *ids_in, size_t k0)
{
    for (size_t i = 0; i < k0; i++)
        maxheap_push(i + 1, vals, ids, x[i], ids_in ? ids_in[i] : (long)i);
    for (size_t i = k0; i < k; i++) {
        vals[i] = FLT_MAX;
        ids[i] = -1;
    }
}

void maxheap_pop(size_t k, float *vals, long *ids)
{
    // 1-based sift-down of the last element from the root
    float *v = vals - 1;
    long *id = ids - 1;
    const float last = v[k];
    size_t hole = 1;
    for (;;) {
        const size_t l = hole * 2, r = l + 1;
        if (l > k)
            break;
        const size_t big = (r == k + 1 || v[l] > v[r]) ? l : r;
        if (last > v[big])
            break;
        v[hole] = v[big];
        id[hole] = id[big];
        hole = big;
    }
    v[hole] = v[k];
    id[hole] = id[k];
}

void maxheap_push(size_t k, float *vals, long *ids, float nv, long nid)
{
    float *v = vals - 1;
    long *id = ids - 1;
    size_t hole = k;
    while (hole > 1) {
        const size_t parent = hole / 2;
        if (!(nv > v[parent]))
            break;
        v[hole] = v[parent];
        id[hole] = id[parent];
        hole = parent;
    }
    v[hole] = nv;
    id[hole] = nid;
}

// ---------------------------------------------------------------------------------------------- utils
// SSE order of faiss's fvec_inner_product: 4 partial sums over blocks of 4, zero-padded tail, (s0+s1)+(s2+s3)
float fvec_inner_product(const float *x, const float *y, size_t d)
{
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int l = 0; l < 4; l++)
            s[l] = s[l] + x[i + l] * y[i + l];
    for (int l = 0; i + l < d; l++)
        s[l] = s[l] + x[i + l] * y[i + l];
    return (s[0] + s[1]) + (s[2] + s[3]);
}

float fvec_norm_L2sqr(const float *x, size_t d) { return fvec_inner_product(x, x, d); }

void fvec_norms_L2sqr(float *nr, const float *x, size_t d, size_t nx)
{
    for (size_t i = 0; i < nx; i++)
        nr[i] = fvec_norm_L2sqr(x + i * d, d);
}

void fvec_madd(size_t n, const float *a, float bf, const float *b, float *c)
{
    for (size_t i = 0; i < n; i++)
        c[i] = a[i] + bf * b[i];
}

void rand_perm(int *perm, size_t n, long seed)
{
    for (size_t i = 0; i < n; i++)
        perm[i] = (int)i;
    std::mt19937 rng((unsigned)seed);
    for (size_t i = 0; i + 1 < n; i++) {
        const size_t j = i + rng() % (n - i);
        std::swap(perm[i], perm[j]);
    }
}

// ---------------------------------------------------------------------------------------------- PQ
ProductQuantizer::ProductQuantizer(size_t d_, size_t M_, size_t nbits_)
    : d(d_), M(M_), nbits(nbits_), dsub(M_ ? d_ / M_ : 0), byte_per_idx((nbits_ + 7) / 8),
      code_size(M_ * ((nbits_ + 7) / 8)), ksub((size_t)1 << nbits_), verbose(false)
{
    if (M == 0 || d % M)
        throw std::runtime_error("ProductQuantizer: d must be a multiple of M");
    if (nbits != 8)
        throw std::runtime_error("ProductQuantizer: only nbits = 8 is supported");
    centroids.assign(d * ksub, 0.f);
}

ProductQuantizer::ProductQuantizer() : ProductQuantizer(1, 1, 8) {}

void ProductQuantizer::compute_inner_prod_table(const float *x, float *dis_table) const
{
    for (size_t m = 0; m < M; m++)
        for (size_t c = 0; c < ksub; c++)
            dis_table[m * ksub + c] = fvec_inner_product(x + m * dsub, get_centroids(m, c), dsub);
}

void ProductQuantizer::decode(const uint8_t *code, float *x) const
{
    for (size_t m = 0; m < M; m++)
        std::memcpy(x + m * dsub, get_centroids(m, code[m]), dsub * sizeof(float));
}

void ProductQuantizer::decode(const uint8_t *code, float *x, size_t n) const
{
    for (size_t i = 0; i < n; i++)
        decode(code + i * code_size, x + i * d);
}

// SSE order of faiss's fvec_L2sqr: 4 partial sums over blocks of 4, zero-padded tail, (s0+s1)+(s2+s3)
float fvec_L2sqr(const float *x, const float *y, size_t d)
{
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int l = 0; l < 4; l++) {
            const float t = x[i + l] - y[i + l];
            s[l] = s[l] + t * t;
        }
    for (int l = 0; i + l < d; l++) {
        const float t = x[i + l] - y[i + l];
        s[l] = s[l] + t * t;
    }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

// faiss: first code word with dis < mindis, the search starting at mindis = 1e20 and index -1
void ProductQuantizer::compute_code(const float *x, uint8_t *code) const
{
    for (size_t m = 0; m < M; m++) {
        float best = 1e20f;
        int arg = -1;
        for (size_t c = 0; c < ksub; c++) {
            const float dist = fvec_L2sqr(x + m * dsub, get_centroids(m, c), dsub);
            if (dist < best) {
                best = dist;
                arg = (int)c;
            }
        }
        code[m] = (uint8_t)arg;
    }
}

void ProductQuantizer::compute_codes(const float *x, uint8_t *codes, size_t n) const
{
#pragma omp parallel for
    for (long i = 0; i < (long)n; i++)
        compute_code(x + (size_t)i * d, codes + (size_t)i * code_size);
}

void ProductQuantizer::train(int n, const float *x) { train_iters(n, x, 25, false); }

void ProductQuantizer::train_iters(int n, const float *x, int niter, bool warm)
{
    // Lloyd iterations per sub-space, seeded with a random subset; the iterations themselves run on the device
    // (assignment = compute_codes with the current code book, update = means in point order: ivfhnsw_gpu_pq_train)
    if (!warm) {
        std::vector<int> perm((size_t)n);
        for (size_t m = 0; m < M; m++) {
            rand_perm(perm.data(), (size_t)n, 1234 + (long)m);
            for (size_t c = 0; c < ksub; c++)
                std::memcpy(get_centroids(m, c), x + (size_t)perm[c % (size_t)n] * d + m * dsub, dsub * sizeof(float));
        }
    }
    TrainDevice dev;
    if (ivfhnsw_gpu_pq_train(dev.h, (size_t)n, d, M, x, (size_t)niter, centroids.data(), nullptr))
        throw std::runtime_error(std::string("ivfhnsw_gpu_pq_train: ") + ivfhnsw_gpu_last_error());
    if (verbose)
        printf("  PQ: %zu sub-quantizers trained, %d iterations\n", M, niter);
}

// ---------------------------------------------------------------------------------------------- transforms
void VectorTransform::train(long, const float *) {}

float *VectorTransform::apply(long n, const float *x) const
{
    float *xt = new float[(size_t)n * d_out];
    apply_noalloc(n, x, xt);
    return xt;
}

LinearTransform::LinearTransform(int d_in_, int d_out_, bool have_bias_)
    : VectorTransform(d_in_, d_out_), have_bias(have_bias_), verbose(false)
{
}

// y[i] = fmaf chain over k of A[i][k] * x[k], then + b[i]: the order the device kernel and the oracle use
// (faiss calls sgemm here; its summation order is unspecified)
void LinearTransform::apply_noalloc(long n, const float *x, float *xt) const
{
    if (A.size() != (size_t)d_in * d_out)
        throw std::runtime_error("LinearTransform: matrix not initialised");
    for (long q = 0; q < n; q++)
        for (int i = 0; i < d_out; i++) {
            float acc = 0.f;
            for (int k = 0; k < d_in; k++)
                acc = std::fmaf(A[(size_t)i * d_in + k], x[(size_t)q * d_in + k], acc);
            xt[(size_t)q * d_out + i] = have_bias ? acc + b[i] : acc;
        }
}

void LinearTransform::transform_transpose(long n, const float *y, float *x) const
{
    for (long q = 0; q < n; q++)
        for (int k = 0; k < d_in; k++) {
            float acc = 0.f;
            for (int i = 0; i < d_out; i++) {
                const float yi = have_bias ? y[(size_t)q * d_out + i] - b[i] : y[(size_t)q * d_out + i];
                acc = std::fmaf(A[(size_t)i * d_in + k], yi, acc);
            }
            x[(size_t)q * d_in + k] = acc;
        }
}

OPQMatrix::OPQMatrix(int d, int M_, int d2)
    : LinearTransform(d, d2 == -1 ? d : d2, false), M(M_), niter(50), niter_pq(4), niter_pq_0(40),
      max_train_points(256 * 256)
{
    is_trained = false;
}

// One-sided Jacobi SVD of the square matrix c (row major, n x n, overwritten): on return the columns of c are
// u_j * s_j and v holds V (row major), c = U S V^T.  Plain, slow and enough for d x d with d in the hundreds.
static void jacobi_svd(std::vector<double> &c, std::vector<double> &v, int n)
{
    v.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < n; i++) {
                    const double x = c[(size_t)i * n + p], y = c[(size_t)i * n + q];
                    a += x * x;
                    b += y * y;
                    g += x * y;
                }
                if (std::fabs(g) <= 1e-15 * std::sqrt(a * b))
                    continue;
                off += std::fabs(g);
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; i++) {
                    const double x = c[(size_t)i * n + p], y = c[(size_t)i * n + q];
                    c[(size_t)i * n + p] = cs * x - sn * y;
                    c[(size_t)i * n + q] = sn * x + cs * y;
                    const double vx = v[(size_t)i * n + p], vy = v[(size_t)i * n + q];
                    v[(size_t)i * n + p] = cs * vx - sn * vy;
                    v[(size_t)i * n + q] = sn * vx + cs * vy;
                }
            }
        if (off < 1e-12)
            break;
    }
}

// Non-parametric OPQ (Ge, He, Ke, Sun 2013), the scheme faiss's OPQMatrix::train follows: start from a random
// rotation, then alternate (a) product quantizer training on the rotated points (niter_pq_0 Lloyd iterations the
// first time, niter_pq warm-started ones after) and (b) the orthogonal Procrustes step -- the rotation R that
// minimises sum ||R x_i - y_i||^2 against the decoded points y_i is V U^T for X^T Y = U S V^T.
// Construction side, host only; faiss's own random streams are not reproduced, so matrices differ from faiss's.
void OPQMatrix::train(long n_in, const float *x)
{
    const int d = d_in;
    if (d_out != d_in)
        throw std::runtime_error("OPQMatrix::train: only square rotations (d_out == d_in) are supported");
    if (M <= 0 || d % M)
        throw std::runtime_error("OPQMatrix::train: d must be a multiple of M");
    const size_t n = (size_t)std::min<long>(n_in, (long)max_train_points);
    if (n < 256)
        throw std::runtime_error("OPQMatrix::train: needs at least 256 training points");
    // random orthonormal start: Gram-Schmidt of a seeded Gaussian matrix
    A.assign((size_t)d * d, 0.f);
    {
        std::mt19937 rng(1234);
        std::normal_distribution<double> gauss(0.0, 1.0);
        std::vector<double> q((size_t)d * d);
        for (auto &e : q)
            e = gauss(rng);
        for (int i = 0; i < d; i++) {
            for (int j = 0; j < i; j++) {
                double dot = 0;
                for (int k = 0; k < d; k++)
                    dot += q[(size_t)i * d + k] * q[(size_t)j * d + k];
                for (int k = 0; k < d; k++)
                    q[(size_t)i * d + k] -= dot * q[(size_t)j * d + k];
            }
            double nrm = 0;
            for (int k = 0; k < d; k++)
                nrm += q[(size_t)i * d + k] * q[(size_t)i * d + k];
            nrm = std::sqrt(nrm);
            for (int k = 0; k < d; k++)
                q[(size_t)i * d + k] /= nrm;
        }
        for (size_t e = 0; e < q.size(); e++)
            A[e] = (float)q[e];
    }
    have_bias = false;
    b.clear();
    ProductQuantizer pq((size_t)d, (size_t)M, 8);
    std::vector<float> xr(n * d), xd(n * d);
    std::vector<uint8_t> codes(n * pq.code_size);
    std::vector<double> c((size_t)d * d), v;
    for (int it = 0; it < niter; it++) {
        apply_noalloc((long)n, x, xr.data());
        pq.train_iters((int)n, xr.data(), it == 0 ? niter_pq_0 : niter_pq, it != 0);
        pq.compute_codes(xr.data(), codes.data(), n);
        pq.decode(codes.data(), xd.data(), n);
        if (verbose) {
            double err = 0;
            for (size_t e = 0; e < n * (size_t)d; e++)
                err += (double)(xr[e] - xd[e]) * (xr[e] - xd[e]);
            printf("  OPQ iteration %d/%d: quantisation error %.6g\n", it + 1, niter, err / (double)n);
        }
        // C = X^T Y (d x d), C[a][b] = sum_i x_i[a] * y_i[b]: on the matrix cores (ivfhnsw_gpu_xty)
        {
            TrainDevice dev;
            std::vector<float> cf((size_t)d * d);
            if (ivfhnsw_gpu_xty(dev.h, n, (size_t)d, x, xd.data(), cf.data()))
                throw std::runtime_error(std::string("ivfhnsw_gpu_xty: ") + ivfhnsw_gpu_last_error());
            for (size_t e = 0; e < cf.size(); e++)
                c[e] = (double)cf[e];
        }
        jacobi_svd(c, v, d); // columns of c: u_j * s_j
        // R = V U^T: R[i][k] = sum_j V[i][j] * U[k][j]
        for (int j = 0; j < d; j++) {
            double s = 0;
            for (int k = 0; k < d; k++)
                s += c[(size_t)k * d + j] * c[(size_t)k * d + j];
            s = std::sqrt(s);
            if (s > 0)
                for (int k = 0; k < d; k++)
                    c[(size_t)k * d + j] /= s;
        }
        for (int i = 0; i < d; i++)
            for (int k = 0; k < d; k++) {
                double r = 0;
                for (int j = 0; j < d; j++)
                    r += v[(size_t)i * d + j] * c[(size_t)k * d + j];
                A[(size_t)i * d + k] = (float)r;
            }
    }
    is_trained = true;
}

// ---------------------------------------------------------------------------------------------- file I/O
namespace {

struct File {
    FILE *f;
    std::string name;
    File(const char *fname, const char *mode) : f(fopen(fname, mode)), name(fname)
    {
        if (!f)
            throw std::runtime_error("cannot open " + name);
    }
    ~File() { fclose(f); }
    void rd(void *p, size_t sz, size_t n)
    {
        if (fread(p, sz, n, f) != n)
            throw std::runtime_error("short read in " + name);
    }
    void wr(const void *p, size_t sz, size_t n)
    {
        if (fwrite(p, sz, n, f) != n)
            throw std::runtime_error("short write in " + name);
    }
    void rdvec(std::vector<float> &v, size_t max_elems)
    {
        size_t n = 0;
        rd(&n, sizeof(n), 1);
        if (n > max_elems)
            throw std::runtime_error("implausible vector length in " + name);
        v.resize(n);
        rd(v.data(), sizeof(float), n);
    }
    void wrvec(const std::vector<float> &v)
    {
        const size_t n = v.size();
        wr(&n, sizeof(n), 1);
        wr(v.data(), sizeof(float), n);
    }
};

const uint32_t kFourccLTra = (uint32_t)'L' | ((uint32_t)'T' << 8) | ((uint32_t)'r' << 16) | ((uint32_t)'a' << 24);

} // namespace

void write_ProductQuantizer(const ProductQuantizer *pq, const char *fname)
{
    File f(fname, "wb");
    f.wr(&pq->d, sizeof(size_t), 1);
    f.wr(&pq->M, sizeof(size_t), 1);
    f.wr(&pq->nbits, sizeof(size_t), 1);
    f.wrvec(pq->centroids);
}

ProductQuantizer *read_ProductQuantizer(const char *fname)
{
    File f(fname, "rb");
    size_t d, M, nbits;
    f.rd(&d, sizeof(size_t), 1);
    f.rd(&M, sizeof(size_t), 1);
    f.rd(&nbits, sizeof(size_t), 1);
    if (d == 0 || M == 0 || d % M || nbits != 8 || d > (1u << 20))
        throw std::runtime_error(std::string("bad ProductQuantizer header in ") + fname);
    ProductQuantizer *pq = new ProductQuantizer(d, M, nbits);
    f.rdvec(pq->centroids, d * pq->ksub);
    if (pq->centroids.size() != d * pq->ksub) {
        delete pq;
        throw std::runtime_error(std::string("ProductQuantizer centroid count != d * ksub in ") + fname);
    }
    return pq;
}

void write_VectorTransform(const VectorTransform *vt, const char *fname)
{
    const LinearTransform *lt = dynamic_cast<const LinearTransform *>(vt);
    if (!lt)
        throw std::runtime_error("write_VectorTransform: only LinearTransform / OPQMatrix are supported");
    File f(fname, "wb");
    f.wr(&kFourccLTra, sizeof(uint32_t), 1);
    f.wr(&lt->have_bias, sizeof(bool), 1);
    f.wrvec(lt->A);
    f.wrvec(lt->b);
    f.wr(&vt->d_in, sizeof(int), 1);
    f.wr(&vt->d_out, sizeof(int), 1);
    f.wr(&vt->is_trained, sizeof(bool), 1);
}

VectorTransform *read_VectorTransform(const char *fname)
{
    File f(fname, "rb");
    uint32_t h = 0;
    f.rd(&h, sizeof(h), 1);
    if (h != kFourccLTra)
        throw std::runtime_error(std::string("not a LinearTransform ('LTra') file: ") + fname);
    LinearTransform *lt = new LinearTransform();
    try {
        f.rd(&lt->have_bias, sizeof(bool), 1);
        f.rdvec(lt->A, (size_t)1 << 32);
        f.rdvec(lt->b, (size_t)1 << 20);
        f.rd(&lt->d_in, sizeof(int), 1);
        f.rd(&lt->d_out, sizeof(int), 1);
        f.rd(&lt->is_trained, sizeof(bool), 1);
        if (lt->d_in <= 0 || lt->d_out <= 0 || lt->A.size() != (size_t)lt->d_in * lt->d_out ||
            (lt->have_bias && lt->b.size() != (size_t)lt->d_out))
            throw std::runtime_error(std::string("inconsistent LinearTransform in ") + fname);
    } catch (...) {
        delete lt;
        throw;
    }
    return lt;
}

} // namespace faiss
