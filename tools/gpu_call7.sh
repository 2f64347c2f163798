#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02g; mkdir -p $O
R=$ROOT/tools/bin/rocfft_repro
run() { timeout -k 5 120 $R "$@" 2>&1 | grep -E "sequence|CHECK|FAILED"; }
{
run =G:16x1x1:1 =H:32x1x1:1 cG cH xG xH
run =G:16x4x1:1 =H:32x4x1:1 cG cH xG xH
run =G:16x8x4:2 =H:32x8x4:2 cG cH xG xH
run =G:16x16x4:2 =H:32x8x4:2 cG cH xG xH
run =G:16x16x16:3 =H:32x16x4:2 cG cH xG xH
run =G:16x16x16:3 =H:32x8x4:2 cG cH xG xH
run =G:16x16x16:3 =H:64x8x4:2 cG cH xG xH
run =G:16x16x16:3 =H:32x8x1:2 cG cH xG xH
run =G:16x16x4:2 =H:32x8x8:2 cG cH xG xH
run =G:8x16x4:2 =H:32x8x8:2 cG cH xG xH
run =G:48x48x4:2 =H:96x24x4:2 cG cH xG xH
run =G:64x64x64:3 =H:128x32x4:2 cG cH xG xH
run =G:16x16x16:3 =H:32x8x8:2 =I:16x16x16:3 cG cH cI xG xH xI
} | tee $O/rocfft_repro3.log
