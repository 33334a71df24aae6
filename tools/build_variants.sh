# Development: A/B builds of the library.  tools/build_variants.sh "tag:hedge flags" ...  (objects start as copies of build/,
# mtimes preserved, so that only what is stale for the tag is recompiled).  FILE=train / FILE=graph builds train.hip / graph.hip with the flags instead.
cd "$(dirname "$0")/.."
for v in "$@"; do
  tag=${v%%:*}; fl=${v#*:}
  mkdir -p gnn_manip_amd/build_$tag
  [ -d gnn_manip_amd/build ] && cp -pn gnn_manip_amd/build/*.o gnn_manip_amd/build_$tag/ 2>/dev/null
  rm -f gnn_manip_amd/build_$tag/${FILE:-hedge}.o
  (case "${FILE:-hedge}" in train) export GM_TRAIN_FLAGS="$fl";; graph) export GM_GRAPH_FLAGS="$fl";; hmlp) export GM_HM_FLAGS="$fl";; *) export GM_HEDGE_FLAGS="$fl";; esac; python -m gnn_manip_amd.build --tag=$tag > /tmp/b_$tag.log 2>&1 && echo "$tag ok" || { echo "$tag FAILED"; tail -5 /tmp/b_$tag.log; }) &
done
wait
