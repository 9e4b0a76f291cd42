#!/bin/bash
# The reference's examples/run_deep1b*.sh, pointed at a synthetic data set (no DEEP1B offline) and at the
# reference's OWN driver binary built unchanged against this repository (make -C oracle ref_drivers).
# Usage: examples/run_synthetic_deep.sh [grouping] [opq]      (needs an MI355X)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
DATA="${TMPDIR:-/tmp}/ivfhnsw_synth_$$"
GROUPING=""; OPQ="off"; DRIVER="test_ivfhnsw_deep1b"; EXTRA=""
for a in "$@"; do
  [ "$a" = "grouping" ] && { GROUPING="--grouping"; DRIVER="test_ivfhnsw_grouping_deep1b"; EXTRA="-nsubc 16 -pruning on"; }
  [ "$a" = "opq" ] && OPQ="on"
done
[ "$OPQ" = "on" ] && OPQFLAG="--opq" || OPQFLAG=""
python3 "$ROOT/examples/make_synthetic_dataset.py" "$DATA" $GROUPING $OPQFLAG --nc 1024 --nb 100000 --nq 1000 --d 96

# same flags as the reference's run scripts
"$ROOT/oracle/_ref/$DRIVER" \
  -M 16 -efConstruction 200 -nb 100000 -nt 10000 -nsubt 10000 -nc 1024 -nq 1000 -ngt 1 -d 96 \
  -code_size 16 -opq $OPQ -k 1 -nprobe 32 -max_codes 10000 -efSearch 80 $EXTRA \
  -path_base unused -path_learn unused -path_q "$DATA/queries.fvecs" -path_gt "$DATA/groundtruth.ivecs" \
  -path_centroids "$DATA/centroids.fvecs" -path_precomputed_idx "$DATA/precomputed_idxs.ivecs" \
  -path_info "$DATA/hnsw.info" -path_edges "$DATA/hnsw.edges" -path_pq "$DATA/pq.dat" \
  -path_opq_matrix "$DATA/opq.dat" -path_norm_pq "$DATA/norm_pq.dat" -path_index "$DATA/corpus.index"
rm -rf "$DATA"
