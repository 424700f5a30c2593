# Runs ON THE GPU BOX: quick look at the large-K configs (no-replan and replan)
cd $GRAFT_REPO_ROOT
source tools/ab_large_k.sh.inc
for cfg in "56403 1280 8 0.2" "27000 65504 1 0.1" "50000 1280 16 0.06" "20000 1280 64 0.1"; do set -- $cfg
  echo "== K=$1 T=$2 blocks=$3 loss=$4"; echo "  no-replan:"; run $1 $2 $3 $4 --no-replan; echo "  replan:"; run $1 $2 $3 $4
done
