// Command-line options of the reference's drivers (reference Parser.h): the same public fields and the same
// "-flag value" syntax, so examples/run_*.sh keep working.  Table driven.  As in the reference there are no
// defaults, "-h"/"--help"/no arguments print the usage and exit, a trailing flag without a value is ignored, and
// the precomputed-index flag is spelled "-path_precomputed_idx" (Parser.h:123; the accepted alias
// "-path_precomputed_idxs" is what the usage text has always advertised).
#ifndef IVFHNSW_AMD_PARSER_H
#define IVFHNSW_AMD_PARSER_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

struct Parser {
    const char *cmd;

    // HNSW
    size_t M;
    size_t efConstruction;
    // data
    size_t nb;
    size_t nt;
    size_t nsubt;
    size_t nc;
    size_t nsubc;
    size_t nq;
    size_t ngt;
    size_t d;
    // PQ
    size_t code_size;
    bool do_opq;
    // search
    size_t k;
    size_t nprobe;
    size_t max_codes;
    size_t efSearch;
    bool do_pruning;
    // paths
    const char *path_base;
    const char *path_learn;
    const char *path_q;
    const char *path_gt;
    const char *path_centroids;
    const char *path_precomputed_idxs;
    const char *path_info;
    const char *path_edges;
    const char *path_pq;
    const char *path_opq_matrix;
    const char *path_norm_pq;
    const char *path_index;

    Parser(int argc, char **argv)
    {
        cmd = argv[0];
        if (argc == 1)
            usage();

        struct NumOpt { const char *flag; size_t Parser::*field; const char *help; };
        struct SwitchOpt { const char *flag; bool Parser::*field; const char *help; };
        struct PathOpt { const char *flag; const char *Parser::*field; const char *help; };
        static const NumOpt nums[] = {
            {"-M", &Parser::M, nullptr}, {"-efConstruction", &Parser::efConstruction, nullptr},
            {"-nb", &Parser::nb, nullptr}, {"-nc", &Parser::nc, nullptr}, {"-nsubc", &Parser::nsubc, nullptr},
            {"-nt", &Parser::nt, nullptr}, {"-nsubt", &Parser::nsubt, nullptr}, {"-nq", &Parser::nq, nullptr},
            {"-ngt", &Parser::ngt, nullptr}, {"-d", &Parser::d, nullptr},
            {"-code_size", &Parser::code_size, nullptr}, {"-k", &Parser::k, nullptr},
            {"-nprobe", &Parser::nprobe, nullptr}, {"-max_codes", &Parser::max_codes, nullptr},
            {"-efSearch", &Parser::efSearch, nullptr}};
        static const SwitchOpt switches[] = {{"-opq", &Parser::do_opq, nullptr}, {"-pruning", &Parser::do_pruning, nullptr}};
        static const PathOpt paths[] = {
            {"-path_base", &Parser::path_base, nullptr}, {"-path_learn", &Parser::path_learn, nullptr},
            {"-path_q", &Parser::path_q, nullptr}, {"-path_gt", &Parser::path_gt, nullptr},
            {"-path_centroids", &Parser::path_centroids, nullptr},
            {"-path_precomputed_idx", &Parser::path_precomputed_idxs, nullptr},
            {"-path_precomputed_idxs", &Parser::path_precomputed_idxs, nullptr},
            {"-path_info", &Parser::path_info, nullptr}, {"-path_edges", &Parser::path_edges, nullptr},
            {"-path_pq", &Parser::path_pq, nullptr}, {"-path_opq_matrix", &Parser::path_opq_matrix, nullptr},
            {"-path_norm_pq", &Parser::path_norm_pq, nullptr}, {"-path_index", &Parser::path_index, nullptr}};

        for (int i = 1; i < argc; i++) {
            const char *a = argv[i];
            if (!strcmp(a, "-h") || !strcmp(a, "--help"))
                usage();
            if (i == argc - 1)
                break; // a last flag has no value
            bool hit = false;
            for (const NumOpt &o : nums)
                if (!hit && !strcmp(a, o.flag)) {
                    sscanf(argv[++i], "%zu", &(this->*o.field));
                    hit = true;
                }
            for (const SwitchOpt &o : switches)
                if (!hit && !strcmp(a, o.flag)) {
                    this->*o.field = !strcmp(argv[++i], "on");
                    hit = true;
                }
            for (const PathOpt &o : paths)
                if (!hit && !strcmp(a, o.flag)) {
                    this->*o.field = argv[++i];
                    hit = true;
                }
        }
    }

    void usage()
    {
        printf("Usage: %s [options]\n", cmd);
        printf("HNSW:    -M #  -efConstruction #\n"
               "Data:    -nb #  -nt #  -nsubt #  -nc #  -nsubc #  -nq #  -ngt #  -d #\n"
               "PQ:      -code_size #  -opq on/off\n"
               "Search:  -k #  -nprobe #  -max_codes #  -efSearch #  -pruning on/off\n"
               "Paths:   -path_base  -path_learn  -path_q  -path_gt  -path_centroids  -path_precomputed_idx\n"
               "         -path_info  -path_edges  -path_pq  -path_opq_matrix  -path_norm_pq  -path_index\n"
               "(flag meanings as in the reference's examples/run_*.sh)\n");
        exit(0);
    }
};

#endif
