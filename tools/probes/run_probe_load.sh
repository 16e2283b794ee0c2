#!/bin/bash
# which packed-fp32 instruction forms go wrong while other waves of the SIMD run MFMAs (see probe_load_after_mfma.hip)
B="hipcc -O3 -Wno-unused-value --offload-arch=gfx950 tools/probes/probe_load_after_mfma.hip -o /tmp/probe_load"
run() { echo "-- $1"; shift; "$@" 2>/dev/null && /tmp/probe_load 40 0 | sed 's/.*wrong by lane quarter 0..3 //; s/| copy.*//'; }
run "1 v_pk_mul_f32 op_sel_hi:[0,1]" $B '-DFIRST_USE="v_pk_mul_f32 %1, %0, %4 op_sel_hi:[0,1]"' '-DWANT_LO=(want*3.0f)' '-DWANT_HI=(want*5.0f)'
run "2 v_pk_fma_f32 op_sel_hi:[1,1,0]" $B '-DFIRST_USE="v_pk_fma_f32 %1, %0, %4, %5 op_sel_hi:[1,1,0]"' '-DWANT_LO=(want*3.0f)' '-DWANT_HI=((want+0.25f)*5.0f)'
run "3 v_pk_add_f32 op_sel_hi:[1,0]" $B '-DFIRST_USE="v_pk_add_f32 %1, %0, %4 op_sel_hi:[1,0]"' '-DWANT_LO=(want+3.0f)' '-DWANT_HI=((want+0.25f)+3.0f)'
run "4 v_pk_fma_f32 op_sel_hi:[0,1,1]" $B '-DFIRST_USE="v_pk_fma_f32 %1, %0, %4, %5 op_sel_hi:[0,1,1]"' '-DWANT_LO=(want*3.0f)' '-DWANT_HI=(want*5.0f)'
run "5 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]" $B
run "5b the same with the loaded pair as src1: v_pk_mul_f32 x, d op_sel:[0,1] op_sel_hi:[0,0]" $B '-DFIRST_USE="v_pk_mul_f32 %1, %4, %0 op_sel:[0,1] op_sel_hi:[0,0]"' '-DWANT_LO=(3.0f*(want+0.25f))' '-DWANT_HI=(3.0f*want)'
run "6 v_pk_fma_f32 op_sel:[1,0,0]" $B '-DFIRST_USE="v_pk_fma_f32 %1, %0, %4, %5 op_sel:[1,0,0]"' '-DWANT_LO=((want+0.25f)*3.0f)' '-DWANT_HI=((want+0.25f)*5.0f)'
run "7 v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0]" $B '-DFIRST_USE="v_pk_mul_f32 %1, %0, %4 op_sel:[1,1] op_sel_hi:[1,0]"' '-DWANT_LO=((want+0.25f)*5.0f)' '-DWANT_HI=((want+0.25f)*3.0f)'
run "8 v_pk_fma_f32 op_sel_hi:[1,0,0]" $B '-DFIRST_USE="v_pk_fma_f32 %1, %0, %4, %5 op_sel_hi:[1,0,0]"' '-DWANT_LO=(want*3.0f)' '-DWANT_HI=((want+0.25f)*3.0f)'
run "9 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,0]" $B '-DFIRST_USE="v_pk_add_f32 %1, %0, %4 op_sel:[0,1] op_sel_hi:[0,0]"' '-DWANT_LO=(want+5.0f)' '-DWANT_HI=(want+3.0f)'
run "10 v_pk_mul_f32 op_sel:[0,1] (hi default)" $B '-DFIRST_USE="v_pk_mul_f32 %1, %0, %4 op_sel:[0,1]"' '-DWANT_LO=(want*5.0f)' '-DWANT_HI=((want+0.25f)*5.0f)'
run "11 v_pk_mul_f32 op_sel:[1,0]" $B '-DFIRST_USE="v_pk_mul_f32 %1, %0, %4 op_sel:[1,0]"' '-DWANT_LO=((want+0.25f)*3.0f)' '-DWANT_HI=((want+0.25f)*5.0f)'
run "12 v_pk_mov_b32 d, d op_sel:[1,0] (the skinny GEMM kernel's epilogue; only the LOW result is checked)" $B '-DFIRST_USE="v_pk_mov_b32 %1, %0, %0 op_sel:[1,0]"' '-DWANT_LO=(want+0.25f)' '-DWANT_HI=(prod[1])'
run "13 v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[0,0,0]" $B '-DFIRST_USE="v_pk_fma_f32 %1, %0, %4, %5 op_sel:[0,1,0] op_sel_hi:[0,0,0]"' '-DWANT_LO=(want*5.0f)' '-DWANT_HI=(want*3.0f)'
run "14 v_pk_fma_f32 x, x, d op_sel:[0,0,1] (src2 high half for the low result)" $B '-DFIRST_USE="v_pk_fma_f32 %1, %4, %4, %0 op_sel:[0,0,1]"' '-DWANT_LO=(9.0f+(want+0.25f))' '-DWANT_HI=(25.0f+(want+0.25f))'
run "15 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] (= form 10)" $B '-DFIRST_USE="v_pk_mul_f32 %1, %0, %4 op_sel:[0,1] op_sel_hi:[1,1]"' '-DWANT_LO=(want*5.0f)' '-DWANT_HI=((want+0.25f)*5.0f)'
for g in "s_nop 3" "s_nop 15" "v_nop\\n\\tv_nop"; do
  run "form 5 after a gap of [$g] behind the s_waitcnt" $B "-DGAP=\"$g\\n\\t\""
done
echo "-- form 5 with ONE block (one wave per SIMD) per CU:"; $B 2>/dev/null && /tmp/probe_load 40 160000 | sed 's/.*wrong by lane quarter 0..3 //; s/| copy.*//'
echo "-- form 5 without MFMAs:"; /tmp/probe_load 0 0 | sed 's/.*wrong by lane quarter 0..3 //; s/| copy.*//'
