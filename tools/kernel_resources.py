"""Print VGPR/SGPR/spill/scratch per kernel from a hipcc -save-temps .s file."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
blocks = re.findall(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', s, re.S)
names = [re.search(r'\.name:\s+(\S+)', b).group(1) for b in blocks]
try:
    dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
except FileNotFoundError:
    dem = names
for b, dn in zip(blocks, dem):
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    dn = dn.replace('pairhmm_hip::(anonymous namespace)::', '').replace('pairhmm_hip::DeviceBatch', 'B').replace('void ', '')
    print(f"{dn:44s} vgpr={g('vgpr_count'):>3s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} spill={g('vgpr_spill_count')} scratch={g('private_segment_fixed_size')} lds={g('group_segment_fixed_size')}")
