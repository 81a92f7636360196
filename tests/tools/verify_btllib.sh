#!/bin/bash
# verify_btllib.sh -- ONE command that checks this repository's ntHash / Bloom-filter arithmetic and .bf file format
# against a REAL btllib installation (the development image has none: hashing + file format are "parity unpinned",
# DESIGN.md section 5).  Run it where btllib is installed, e.g.
#     conda create -n btl -c bioconda -c conda-forge btllib && conda activate btl
#     tests/tools/verify_btllib.sh            # uses $CONDA_PREFIX, or BTLLIB_PREFIX=/path/to/prefix
# It builds two small programs -- dump_btllib.cpp (the calls ntedit.cpp makes: hashing_internals::base_*_hash,
# next_*_hash, canonical, extend_hashes, SEED_TAB / CP_OFF / srol_table, KmerBloomFilter build / save / load /
# contains) and dump_ours.cpp (nte_common.h + bfio.cpp, the arithmetic the HIP kernels run) -- feeds both the same
# fixed sequences and diffs everything:
#   1. seeded, rolled, extended and change-last hashes at k = 25, 32, 33, 64, 127 (ACGT, lower case, N);
#   2. the same on IUPAC / U / other bytes (btllib's behaviour there is not written down anywhere: reported separately);
#   3. a filter built by btllib vs one built by us from the same genome: array bytes identical, either program
#      reads the other's file, contains() agrees for every k-mer of a draft (present and absent ones).
# Exit code 0 = everything identical; the "parity unpinned" notes in DESIGN.md / oracle header can then go.
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
KIT="$HERE/btllib_kit"
PREFIX="${BTLLIB_PREFIX:-${CONDA_PREFIX:-/usr/local}}"
WORK="${1:-$(mktemp -d /tmp/verify_btllib.XXXXXX)}"
CXX="${CXX:-g++}"
mkdir -p "$WORK"
if [ ! -e "$PREFIX/include/btllib/nthash.hpp" ]; then
	echo "verify_btllib: no btllib under $PREFIX (set BTLLIB_PREFIX); nothing verified" >&2
	exit 3
fi
$CXX -O2 -std=c++17 -I"$PREFIX/include" -o "$WORK/dump_btllib" "$KIT/dump_btllib.cpp" -L"$PREFIX/lib" -lbtllib -fopenmp -Wl,-rpath,"$PREFIX/lib" || exit 4
$CXX -O2 -std=c++17 -o "$WORK/dump_ours" "$KIT/dump_ours.cpp" "$HERE/../../ntedit_amd/host/bfio.cpp" "$HERE/../../ntedit_amd/host/params.cpp" || exit 4
python3 "$KIT/make_inputs.py" "$WORK" || exit 4
fail=0
for k in 25 32 33 64 127; do
	"$WORK/dump_btllib" hashes "$WORK/acgt.txt" $k 3 > "$WORK/h_btl_$k.txt"
	"$WORK/dump_ours" hashes "$WORK/acgt.txt" $k 3 > "$WORK/h_our_$k.txt"
	if cmp -s "$WORK/h_btl_$k.txt" "$WORK/h_our_$k.txt"; then echo "hashes  k=$k ACGT/acgt/N : identical ($(wc -l < "$WORK/h_our_$k.txt") lines)"; else echo "hashes  k=$k ACGT/acgt/N : DIFFER (diff $WORK/h_btl_$k.txt $WORK/h_our_$k.txt)"; fail=1; fi
done
"$WORK/dump_btllib" hashes "$WORK/exotic.txt" 25 4 > "$WORK/x_btl.txt"
"$WORK/dump_ours" hashes "$WORK/exotic.txt" 25 4 > "$WORK/x_our.txt"
if cmp -s "$WORK/x_btl.txt" "$WORK/x_our.txt"; then echo "hashes  IUPAC / U / other bytes : identical"; else echo "hashes  IUPAC / U / other bytes : DIFFER (diff $WORK/x_btl.txt $WORK/x_our.txt) -- only drafts with such characters inside hashed k-mers are affected"; fail=1; fi
for spec in "25 3 1048576" "40 4 800003"; do
	set -- $spec
	"$WORK/dump_btllib" build "$WORK/genome.fa" $1 $2 $3 "$WORK/btl_$1.bf" > "$WORK/b_btl_$1.txt"
	"$WORK/dump_ours" build "$WORK/genome.fa" $1 $2 $3 "$WORK/our_$1.bf" > "$WORK/b_our_$1.txt"
	python3 "$KIT/cmp_bf.py" "$WORK/btl_$1.bf" "$WORK/our_$1.bf" || fail=1
	for who in btllib ours; do for file in btl our; do
		"$WORK/dump_$who" query "$WORK/${file}_$1.bf" "$WORK/draft.txt" > "$WORK/q_${who}_${file}_$1.txt" 2>&1
	done; done
	if cmp -s "$WORK/q_btllib_btl_$1.txt" "$WORK/q_ours_btl_$1.txt" && cmp -s "$WORK/q_btllib_our_$1.txt" "$WORK/q_ours_our_$1.txt" && cmp -s "$WORK/q_btllib_btl_$1.txt" "$WORK/q_ours_our_$1.txt"; then
		echo "filter  k=$1 h=$2 $3 B : contains() identical in all four reader x file combinations ($(grep -c '^1' "$WORK/q_ours_our_$1.txt") present, $(grep -c '^0' "$WORK/q_ours_our_$1.txt") absent)"
	else
		echo "filter  k=$1 h=$2 $3 B : contains() DIFFERS (see $WORK/q_*_$1.txt)"; fail=1
	fi
done
if [ $fail = 0 ]; then echo "verify_btllib: ALL IDENTICAL -- hashing and the .bf format are pinned against this btllib"; else echo "verify_btllib: differences found (work dir $WORK)"; fi
exit $fail
