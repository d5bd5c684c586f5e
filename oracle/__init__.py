"""CPU oracle for the Mandelbrot tile path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``distributedmandelbrot_amd/`` does.
"""
