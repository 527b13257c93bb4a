"""Frame-to-frame joint jumps in the packed mocap table (the reference's retargeted clips): IK branch flips that the reference's finite-difference
velocities (ML:48-63) turn into hundreds of rad/s.  A reset that lands on one starts the robot with that joint rate (profiles/r04_nonfinite.txt).
    python tools/scan_mocap_jumps.py [threshold_rad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from lifelike_agility_and_play_amd import mocap

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
t = mocap.load_mocap('', 0.02)
fs = t.frame_step
tot = 0
print('frame step %.6f s, %d clips, %d frames; frame-to-frame joint jumps above %.2f rad (frame, time, largest jump, = rad/s by finite differences):' % (fs, len(t.clip_len), int(t.clip_len.sum()), thr))
for c in range(len(t.clip_len)):
    fr = t.frames[t.clip_off[c]: t.clip_off[c] + t.clip_len[c]]
    dj = np.abs(np.diff(fr[:, 7:19], axis=0)).max(1)
    bad = np.nonzero(dj > thr)[0]
    tot += len(bad)
    if len(bad):
        print('clip %2d %-34s %s' % (c, t.names[c], ', '.join('%d @ %.3f s: %.2f rad = %.0f rad/s' % (b, b * fs, dj[b], dj[b] / fs) for b in bad)))
print('%d jumps in %d frames (%.2g of the frame intervals)' % (tot, int(t.clip_len.sum()), tot / float(t.clip_len.sum())))
