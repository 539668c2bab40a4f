import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.afsk_oracle as O
O._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'libafsk_oracle_asan.so')
O.build = lambda force=False: O._LIB_PATH
import json, numpy as np
from tests.golden_inputs import build_input, build_capture
G=json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'reference_vectors.json')))
n=0
for c in G['decode_cases']:
    x=build_input(c); bits,ci,tf=O.decode_bits(x,c['baud'],c['amp_end']); assert ci==c['clock_idx'] and len(bits)==c['nbits']; n+=1
xs=[build_input(c) for c in G['decode_cases'][:20]]
off=np.cumsum([0]+[len(x) for x in xs[:-1]]).astype(np.int64); ln=np.array([len(x) for x in xs],np.int32); bf=np.array([48000//c['baud'] for c in G['decode_cases'][:20]],np.int32)
O.demod_batch(np.concatenate(xs),off,ln,bf,14000,160,4)
# soft outputs with a margins row that is too short for the stream (writes must stay inside it)
for c in G['decode_cases'][:30]:
    x=build_input(c); r=O.demod_batch_soft(x,[0],[len(x)],[48000//c['baud']],c['amp_end'],out_stride=8,margin_stride=50)
    assert int(r['corrected'][0])==c['soft']['corrected']
for c in G['listen_cases']:
    O.gate_stream(build_capture(c['recipe']),c['amp_start'],c['amp_end'],16)
print("asan run ok", n)
