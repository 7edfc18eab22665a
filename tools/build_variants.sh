#!/bin/bash
# tools/build_variants.sh "tag1:-DFOO=1 -DBAR" "tag2:..." -- experiment builds of liboxcull.so into oxylus_amd/variants/ (in parallel)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/oxylus_amd/csrc" && make > /dev/null || exit 1
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( make variant TAG="$tag" EXTRA="$flags" > /tmp/variant_$tag.log 2>&1 || { echo "variant $tag FAILED"; tail -20 /tmp/variant_$tag.log; } ) &
done
wait
ls -la ../variants/*.so
