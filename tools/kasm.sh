#!/bin/bash
# kasm.sh <file.s> <mangled-kernel-name> : print the ISA of one kernel (label .. .Lfunc_end)
awk -v k="$2" '$0 ~ "^"k":" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' "$1"
