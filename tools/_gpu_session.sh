for i in 1 2 3; do timeout 300 python tools/profile_f.py 2>/dev/null | tail -1; done
timeout 300 python tools/generate_subject.py 2>/dev/null | tail -1
