"""The parts of bench.py (repo root): launch (arguments, rank spawning, the CPU rehearsal), workload (the step being timed),
legs (the timed region's bracket, the extra legs, the CPU baseline), roofline (the dominant kernel's object).  bench.py itself
holds the timed region."""
