// The reference's drivers include <ivf-hnsw/hnswalg.h> (a symlink set up by setup_env_dev.sh:15-49).
#pragma once
#include <hnswlib/hnswalg.h>
