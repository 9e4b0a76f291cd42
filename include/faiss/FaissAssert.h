// Minimal stand-in for the faiss header of the same name: the reference includes it
// (IndexIVF_HNSW.h:13) but uses nothing from it on the search path.
#pragma once
#include <cstdio>
#include <cstdlib>
#define FAISS_ASSERT(X)                                                                  \
    do {                                                                                 \
        if (!(X)) {                                                                      \
            fprintf(stderr, "faiss assertion '%s' failed at %s:%d\n", #X, __FILE__, __LINE__); \
            abort();                                                                     \
        }                                                                                \
    } while (0)
