# share of one chunk a signature needs to become a run (CMS_BA_RUN_MIN_PCT): Schur kernel alone, 16 tracked windows
for pct in ${PCT_SET:-100 75 50 35 25}; do echo "CMS_BA_RUN_MIN_PCT=$pct: $(CMS_BA_RUN_MIN_PCT=$pct timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep -E 'lock-step|signature runs' | cut -c1-150 | tr '\n' ' ')"; done
