#!/bin/bash
B=tools/micro/wave_stream.bin
echo "--- loads only: 1-KB pieces (one per instruction) vs two 512-B pieces of streams 135 MB apart per instruction"
for wpc in 8 16; do $B mode=5 nl=6 cap=6 wpc=$wpc; $B mode=21 nl=12 cap=6 wpc=$wpc; $B mode=21 nl=18 cap=9 wpc=$wpc; done
echo "--- the same with stores (PL-like)"
for wpc in 8; do $B mode=7 nl=6 nst=6 bpl=8 cap=12 wpc=$wpc; $B mode=23 nl=12 nst=6 bpl=8 cap=12 wpc=$wpc; done
