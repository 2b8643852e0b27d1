# sourced by the evidence scripts: OUT=gpurun_out/<tag>, created fresh.  A tag is a round-stamped stage name (r03_a ...);
# re-using one is refused, so a summary can never pick up a file of an earlier run (stale CSV under a fresh title).
tag=$1
case "$tag" in r[0-9][0-9]_*) ;; *) echo "usage: $0 <rNN_stage tag> ...  (e.g. r03_a)" >&2; exit 2;; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$root/gpurun_out/$tag
if [ -e "$OUT/$(basename $0 .sh)" ]; then echo "$OUT/$(basename $0 .sh) exists: pick a new tag" >&2; exit 2; fi
mkdir -p "$OUT/$(basename $0 .sh)"
OUT="$OUT/$(basename $0 .sh)"
export TMPDIR=/tmp
