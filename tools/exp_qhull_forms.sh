#!/bin/bash
# Times the forms of the Qhull replay kernel on batches of several shapes (needs a GPU):
# tools/exp_qhull_forms.sh "<forms>" "<slices>x<points> ..."
set -e
cd "$(dirname "$0")/.."
forms=${1:-"97 98 96 160"}
shapes=${2:-"1000x100000 4000x25000 10000x8192 700x50000 300x50000"}
for sh in $shapes; do
  S=${sh%x*}; n=${sh#*x}
  for f in $forms; do
    printf "%s form %s: " "$sh" "$f"
    PCFIT_QHULL_FORM=$f timeout -k 10 100 python tools/exp_qhull_dev.py "$S" "$n" 2>&1 | tail -1 | cut -c1-9
  done
done
