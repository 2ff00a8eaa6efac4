for i in 1 2 3; do timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
