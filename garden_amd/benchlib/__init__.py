"""Pieces of bench.py (the driver's contract stays there: argument parsing, the timed region, the one JSON line): workloads and
their byte counts, the run's shared state, the exchange step of N > 1, the variants timed beside the headline, the watchdogs. Nothing
here touches oracle/: the CPU baseline and the parity check are bench.py's own legs."""
