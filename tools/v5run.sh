cd /root/repo
for s in odd p32 p40 p40b small; do timeout 120 build/convbench $s 2 41 42 43 39 28; done
for s in l26_3x3 l6_3x3r l23_3x3 l29_3x3 l32_3x3; do timeout 120 build/convbench $s 20 41 43 39 28 1; done
timeout 120 build/convbench l2_3x3 20 42 40 31 24
timeout 120 build/convbench l26_3x3 20 t0 t1 t2
