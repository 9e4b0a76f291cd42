// Host utilities declared in include/ivf-hnsw/utils.h (names and behaviour of the reference's utils.cpp).
#include <ivf-hnsw/utils.h>

#include <algorithm>
#include <cstring>
#include <dirent.h>
#include <sys/stat.h>

namespace ivfhnsw {

void random_subset(const float *x, float *x_out, size_t d, size_t nx, size_t sub_nx)
{
    std::vector<int> perm(nx);
    faiss::rand_perm(perm.data(), nx, 1234); // fixed seed, as utils.cpp:13
    for (size_t i = 0; i < sub_nx; i++)
        std::memcpy(x_out + i * d, x + (size_t)perm[i] * d, sizeof(float) * d);
}

// same arithmetic as HierarchicalNSW::fstdistfunc (utils.cpp:22-52 == hnswalg.cpp:326-357)
float fvec_L2sqr(const float *x, const float *y, size_t d)
{
    typedef float v8f __attribute__((vector_size(32), aligned(4)));
    v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t b = 0; b < 2 * (d >> 4); b++) {
        const v8f diff = *reinterpret_cast<const v8f *>(x + 8 * b) - *reinterpret_cast<const v8f *>(y + 8 * b);
        acc = acc + diff * diff;
    }
    float r = acc[0] + acc[1];
    for (int l = 2; l < 8; l++)
        r = r + acc[l];
    return r;
}

float getL2Distance(const float *query, const char *path_vec, const size_t dim, const long vec_id, vec_t type_v)
{
    std::vector<float> v(dim);
    std::ifstream in(path_vec, std::ios::binary);
    if (type_v == base_vec) { // .bvecs record: uint32 dim + dim bytes
        in.seekg((std::streamoff)vec_id * (std::streamoff)(sizeof(uint32_t) + dim));
        readXvecFvec<uint8_t>(in, v.data(), dim, 1);
    } else if (type_v == centroid_vec) { // .fvecs record
        in.seekg((std::streamoff)vec_id * (std::streamoff)(sizeof(uint32_t) + dim * sizeof(float)));
        readXvec<float>(in, v.data(), dim);
    } else {
        std::cout << "Invalid vector type: " << type_v << std::endl;
        assert(0);
    }
    return fvec_L2sqr(query, v.data(), dim);
}

bool cmp(SearchInfo_t a, SearchInfo_t b)
{
    if (a.distance > b.distance)
        return false;
    if (std::abs(b.distance - a.distance) <= 0.001)
        return a.label < b.label;
    return true;
}

size_t base_vec_num(const char *path_base, size_t vec_dim)
{
    struct stat st;
    if (stat(path_base, &st))
        return 0;
    const size_t rec = sizeof(uint32_t) + vec_dim;
    if ((size_t)st.st_size % rec) {
        std::cout << "Invalid size of file: " << path_base << std::endl;
        assert(0);
    }
    return (size_t)st.st_size / rec;
}

void get_files(const char *path_dir, const char *file_ext, std::vector<std::string> &file_list)
{
    DIR *dir = opendir(path_dir);
    if (!dir) {
        std::cout << "Failed to open dir: " << path_dir << std::endl;
        return;
    }
    const size_t le = strlen(file_ext);
    while (struct dirent *e = readdir(dir)) {
        const char *hit = strstr(e->d_name, file_ext);
        if (hit && strlen(e->d_name) != le && strncmp(hit, file_ext, le) == 0)
            file_list.push_back(e->d_name);
    }
    std::sort(file_list.begin(), file_list.end());
    closedir(dir);
}

void check_files(const char *file_prefix, std::vector<std::string> &file_list)
{
    for (const std::string &s : file_list)
        if (s.compare(0, strlen(file_prefix), file_prefix) != 0)
            assert(0);
}

void get_index_name(const char *path_idx, size_t idx, char *idx_name)
{
    sprintf(idx_name, "%s_%02lu%s", path_idx, (unsigned long)idx, ".index");
}

} // namespace ivfhnsw
