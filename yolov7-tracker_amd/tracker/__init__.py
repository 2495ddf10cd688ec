"""Drop-in mirror of the reference's `tracker/` plugin surface for the SORT / ByteTrack path
(tracker/basetrack.py, tracker/bytetrack.py, tracker/kalman_filter.py, tracker/matching.py),
backed by the device-resident track pool and kernels of liby7t.so."""
