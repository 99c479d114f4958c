#!/bin/bash
# Scratch diagnostic build for tools/archive/back_stamps.py: a STAMPS copy of csrc/ (gym_auv_amd/csrc_stampsb/, git-ignored) in which five stamp
# slots of the search / finish roles are re-used for stamps INSIDE k2_back (after the compaction, after the wait for the beam weights,
# after the free rows, before / after the wave sum).  Patches the copies by exact string replacement: it asserts when the source has moved on.
set -e
cd /root/repo
D=gym_auv_amd/csrc_stampsb && mkdir -p $D && cp gym_auv_amd/csrc/*.hip gym_auv_amd/csrc/*.h gym_auv_amd/csrc/Makefile $D/ && rm -f $D/*.o && python - <<'EOF'
p='gym_auv_amd/csrc_stampsb/k_step_fused.hip'
s=open(p).read()
for slot,txt in ((12,'d.stamps[(size_t)e * 16 + 12] = wall_clock64();'),(5,'d.stamps[(size_t)e * 16 + 5] = wall_clock64();'),(6,'d.stamps[(size_t)e * 16 + 6] = wall_clock64();')):
    assert s.count('if (lane == 0) '+txt)==1, slot
    s=s.replace('if (lane == 0) '+txt,'/*moved*/')
for slot in (10,13):
    import re
    pat = re.compile(r"  if \(live && c == 0\) d\.stamps\[\(size_t\)e \* 16 \+ %d\] = wall_clock64\(\);[^\n]*\n" % slot)
    assert len(pat.findall(s))==1, slot
    s=pat.sub("", s)
open(p,'w').write(s)
p='gym_auv_amd/csrc_stampsb/k2_lidar.hip'
s=open(p).read()
a="    auv_wave_lds_sync();\n    if (!AUV_RUN_L(d, 6)) n_hit = 0;\n"
assert s.count(a)==1
s=s.replace(a,"    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = wall_clock64();\n"+a)
c="    __builtin_amdgcn_s_waitcnt(0x0F70);"
assert s.count(c)==1
s=s.replace(c,"    if (lane == 0) d.stamps[(size_t)e * 16 + 10] = wall_clock64();\n"+c+"\n    if (lane == 0) d.stamps[(size_t)e * 16 + 13] = wall_clock64();")
b="  col = __any(col);\n  if (colav) num = auv_wave_sum(num);\n"
assert s.count(b)==1
s=s.replace(b,"  if (lane == 0) d.stamps[(size_t)e * 16 + 5] = wall_clock64();\n"+b+"  if (lane == 0) d.stamps[(size_t)e * 16 + 6] = wall_clock64();\n")
open(p,'w').write(s)
EOF
make -C $D -j8 STAMPS=1 all 2>&1 | grep -E " error" | head; ls -la $D/libauv_hip.so
