python -m pytest tests -x -q -m gpu 2>&1 | tail -1
ps aux | grep -c python
ps aux | grep python | grep -v grep | cut -c1-150 | head
bash scripts/diag_secondary2.sh
