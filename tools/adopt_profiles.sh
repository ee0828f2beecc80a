#!/bin/bash
# After `gpurun -- 'bash tools/refresh_headline.sh <tag>'` has merged gpurun_out/<tag>/ back: copy the collection into profiles/<tag>_* and the
# traffic files (stamped with the build tag of the library that ran) into profiles/, then check that the tag is the current sources'.
#   usage: tools/adopt_profiles.sh <tag>
set -e
TAG=${1:?usage: tools/adopt_profiles.sh <tag>}
cd "$(dirname "$0")/.."
SRC=gpurun_out/$TAG
for f in "$SRC"/*.json "$SRC"/*.txt; do
  b=$(basename "$f")
  case "$b" in traffic_x2h_*) cp "$f" profiles/"$b" ;; *) cp "$f" profiles/"${TAG}_$b" ;; esac
done
python - <<PY
import json
from targetdiff_amd import build
tag = build.source_tag()
for f in ('traffic_x2h_value.json', 'traffic_x2h_key.json'):
    t = json.load(open('profiles/' + f)).get('build_tag')
    print(f, t, 'OK' if t == tag else f'!= sources {tag}: re-run tools/refresh_headline.sh on this tree')
PY
