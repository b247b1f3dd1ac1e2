#!/bin/bash
B=tools/micro/wave_stream.bin
echo "--- calibration: contiguous per wave"
for wpc in 8 32; do $B mode=10 nst=6 bpl=16 cap=30 wpc=$wpc; $B mode=10 nst=6 bpl=8 cap=30 wpc=$wpc; $B mode=13 nl=6 cap=6 wpc=$wpc; $B mode=15 nl=6 nst=6 bpl=16 cap=6 wpc=$wpc; $B mode=15 nl=9 nst=3 bpl=16 cap=6 wpc=$wpc; done
echo "--- marching pattern: stores only"
for wpc in 8 32; do $B mode=2 nst=6 bpl=8 cap=30 wpc=$wpc;  $B mode=2 nst=6 bpl=16 cap=30 wpc=$wpc; done
echo "--- DMA loads only"
for wpc in 8 16; do for cap in 0 6 12; do $B mode=5 nl=6 cap=$cap wpc=$wpc; done; done
$B mode=5 nl=9 cap=18 wpc=8
$B mode=5 nl=12 cap=12 wpc=8
echo "--- loads + stores (PL-like: 6 DMA + 6 x 8B stores; CORR-like: 9+6 DMA, 6 x 16B)"
for wpc in 8 16; do $B mode=7 nl=6 nst=6 bpl=8 cap=6 wpc=$wpc; $B mode=7 nl=6 nst=6 bpl=8 cap=12 wpc=$wpc; $B mode=7 nl=15 nst=6 bpl=16 cap=6 wpc=$wpc; $B mode=7 nl=9 nst=12 bpl=8 cap=12 wpc=$wpc; done
