#!/bin/bash
python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30
