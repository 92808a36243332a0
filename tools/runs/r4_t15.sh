#!/bin/bash
# phase stamps of the planned units under load (n = 5000) with and without the four-buffer loop
for ring in 1 0; do echo "== ring=$ring"; MADQP_CHOL_MID_RING=$ring tools/mid_probe 5000 | cut -c1-100 | sed -n '2,14p;30,32p'; done
