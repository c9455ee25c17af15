#!/bin/bash
# round 5: cache policy of the cfg3 pair's streams, A/B on one box against the in-tree library:
#   ntx    row pass: non-temporal loads of the signal-spectrum rows          ntst   row pass: non-temporal inter-pass stores
#   ntxst  both                                                              ntcol  column pass: tile-row buffer loads with aux = 2 (nt)
#   ntxcol ntx + ntcol                                                       ntxc   ntx + non-temporal code-spectrum rows
# (variants built by: for v in ...; do tools/build_variant.sh NAME "-DBDS_ROWS_NT_X=1 ..."; done -- see tools/README.md)
VARIANTS="${VARIANTS:-ntx ntst ntxst ntcol ntxcol ntxc}" OUT=r05_nt_ab.txt PRNS=${PRNS:-8} bash tools/exp/r5_ab.sh
