#!/bin/bash
# Writes VERSION (tracked): the commit the working tree is based on + a hash of the kernel sources.  The GPU box receives a snapshot without
# .git, so profile stamps (tools/r03/pmc_json.py) read the commit from this file.  Run before `git commit` of a measured state:
#   bash tools/stamp_version.sh && git add VERSION
cd "$(dirname "$0")/.."
head=$(git rev-parse --short=12 HEAD 2>/dev/null || echo unknown)
src=$(cat samplenerfro_amd/csrc/*.hip samplenerfro_amd/csrc/*.inc samplenerfro_amd/csrc/*.h include/rnerf.h | sha256sum | cut -c1-16)
echo "based-on ${head} csrc-sha16 ${src} $(date -u +%Y-%m-%dT%H:%MZ)" > VERSION
cat VERSION
