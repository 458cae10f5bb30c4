#!/bin/bash
# tests/golden/make_fixtures.sh -- how the data fixtures in this directory were made.
#
# Everything here is DATA the reference repository holds (inputs and expected outputs), copied from
# the read-only checkout; no reference source text is kept.  Files larger than 4 KiB are gzipped.
#   sample-data/*.mtx            <- $REF/sample-data/*.mtx      (Matrix Market inputs; badfile.mtx is empty on purpose)
#   reports/*.txt[.gz]           <- $REF/output-test/*.txt and $REF/build/smvp-toolbox_report_CSR_1619162887.txt
#                                   (the reports the reference's authors committed: timings + y vectors at %g)
# The reference program itself could not be run here to produce more vectors: main-cli.c needs libpopt,
# which the image does not have.
set -eu
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$HERE/sample-data" "$HERE/reports"
for f in "$REF"/sample-data/*.mtx; do
  b=$(basename "$f")
  if [ "$(stat -c %s "$f")" -gt 4096 ]; then gzip -9 -n -c "$f" > "$HERE/sample-data/$b.gz"; else cp "$f" "$HERE/sample-data/$b"; fi
done
for f in "$REF"/output-test/*.txt "$REF"/build/smvp-toolbox_report_*.txt; do
  b=$(basename "$f")
  if [ "$(stat -c %s "$f")" -gt 4096 ]; then gzip -9 -n -c "$f" > "$HERE/reports/$b.gz"; else cp "$f" "$HERE/reports/$b"; fi
done
chmod -R u+w,a-x+X "$HERE/sample-data" "$HERE/reports"
ls -l "$HERE/sample-data" "$HERE/reports"
