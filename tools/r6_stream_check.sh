cd $GRAFT_REPO_ROOT
S='import sys,json; d=json.loads(sys.stdin.read()); s=d["streaming"]; print(sys.argv[1], s["time_to_first_audio_ms"], s["all_chunks_ms"], s["one_shot_host_call_ms"])'
python bench.py --cpu-seconds 1 2>/dev/null | tail -1 | python -c "$S" "default, cpu 1 s:"
VITS_CACHE_MB=100000 python bench.py --cpu-seconds 1 2>/dev/null | tail -1 | python -c "$S" "default, cache 100 GB:"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$S" "no cpu baseline:"
VITS_PERSIST_WHEN=0 python bench.py --cpu-seconds 1 2>/dev/null | tail -1 | python -c "$S" "default, WHEN=0:"
