// Host utilities with the names the reference's drivers use (utils.h:33-161): timer, typed binary I/O for the
// .index files (uint32 count + raw elements), .fvecs/.bvecs/.ivecs readers, the AVX-ordered L2 distance.
#ifndef IVFHNSW_AMD_UTILS_H
#define IVFHNSW_AMD_UTILS_H

#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <queue>
#include <string>
#include <vector>

#include <faiss/utils.h>

#define EPS 0.00001

namespace ivfhnsw {

/// Wall-clock stopwatch (microseconds)
class StopW {
    std::chrono::steady_clock::time_point t0_;

public:
    StopW() : t0_(std::chrono::steady_clock::now()) {}
    float getElapsedTimeMicro()
    {
        return (float)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0_).count();
    }
    void reset() { t0_ = std::chrono::steady_clock::now(); }
};

namespace detail {
// one ".xvecs" record: uint32 dimension, then that many elements of type Disk; `sink(j, value)` receives them
template <typename Disk, typename Sink> inline void read_record(std::istream &in, size_t want_dim, Sink sink)
{
    uint32_t dim = 0;
    in.read(reinterpret_cast<char *>(&dim), sizeof(dim));
    if (dim != want_dim) {
        std::cout << "file error\n";
        exit(1);
    }
    std::vector<Disk> rec(dim);
    in.read(reinterpret_cast<char *>(rec.data()), (std::streamsize)dim * sizeof(Disk));
    for (size_t j = 0; j < dim; j++)
        sink(j, rec[j]);
}
} // namespace detail

/// n records of "uint32 dim, dim elements" (.fvecs / .ivecs / .bvecs), elements kept as stored
template <typename T> void readXvec(std::ifstream &in, T *data, const size_t d, const size_t n = 1)
{
    for (size_t i = 0; i < n; i++)
        detail::read_record<T>(in, d, [&](size_t j, T v) { data[i * d + j] = v; });
}
/// same records, elements converted to float (SIFT bytes -> float)
template <typename T> void readXvecFvec(std::ifstream &in, float *data, const size_t d, const size_t n = 1)
{
    for (size_t i = 0; i < n; i++)
        detail::read_record<T>(in, d, [&](size_t j, T v) { data[i * d + j] = (float)v; });
}
template <typename T> void writeXvec(std::ofstream &out, T *data, const size_t d, const size_t n = 1)
{
    const uint32_t dim = (uint32_t)d;
    for (size_t i = 0; i < n; i++) {
        out.write(reinterpret_cast<const char *>(&dim), sizeof(dim));
        out.write(reinterpret_cast<const char *>(data + i * d), (std::streamsize)d * sizeof(T));
    }
}

/// scalars and vectors of the .index files: a vector is a uint32 element count followed by the elements
template <typename T> void write_variable(std::ostream &out, const T &v)
{
    out.write(reinterpret_cast<const char *>(&v), sizeof(T));
}
template <typename T> void read_variable(std::istream &in, T &v) { in.read(reinterpret_cast<char *>(&v), sizeof(T)); }
template <typename T> void write_vector(std::ostream &out, std::vector<T> &vec)
{
    const uint32_t n = (uint32_t)vec.size();
    write_variable(out, n);
    out.write(reinterpret_cast<const char *>(vec.data()), (std::streamsize)n * sizeof(T));
}
template <typename T> void read_vector(std::istream &in, std::vector<T> &vec)
{
    uint32_t n = 0;
    read_variable(in, n);
    vec.resize(n);
    in.read(reinterpret_cast<char *>(vec.data()), (std::streamsize)n * sizeof(T));
}

inline bool exists(const char *path)
{
    std::ifstream f(path);
    return f.good();
}

enum vec_t { base_vec = 0, centroid_vec = 1 };

typedef struct SearchInfo {
    float distance;
    long label;
} SearchInfo_t;

void random_subset(const float *x, float *x_out, size_t d, size_t nx, size_t sub_nx);
/// 8-accumulator L2 over blocks of 16 floats (utils.cpp:22-52): the arithmetic every coarse distance uses
float fvec_L2sqr(const float *x, const float *y, size_t d);
float getL2Distance(const float *query, const char *path_base, const size_t dim, const long vec_id, vec_t type_v);
bool cmp(SearchInfo_t a, SearchInfo_t b);
size_t base_vec_num(const char *path_base, size_t vec_dim);
void get_files(const char *path_dir, const char *file_ext, std::vector<std::string> &file_list);
void check_files(const char *file_prefix, std::vector<std::string> &file_list);
void get_index_name(const char *path_idx, size_t idx, char *idx_name);

} // namespace ivfhnsw
#endif
