"""CPU oracle for the PAniC-3D triplane rendering hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (panic3d-anime-reconstruction_amd / panic3d_amd) never does.
"""
