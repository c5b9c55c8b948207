for la in 0 1 2 8; do
  echo "CMS_BA_LEFTOVER_LOOKAHEAD=$la: $(CMS_BA_LEFTOVER_LOOKAHEAD=$la timeout 300 python tools/prof_ba_many.py 16 track diff 3 2>&1 | grep 'lock-step' | cut -c1-140)"
  CMS_BA_LEFTOVER_LOOKAHEAD=$la CMS_BA_CREATE_TIMING=1 python tools/prof_ba_create.py 8 2>&1 | grep "cms_ba_create\]" | tail -2 | cut -c1-200
done
