python tools/exp_stream_priority.py
CRD_LATE_PRIO=1 python tools/exp_stream_priority.py
CRD_MAIN_PRIO=-1 python tools/exp_stream_priority.py
CRD_MAIN_PRIO=-1 CRD_LATE_PRIO=1 python tools/exp_stream_priority.py
CRD_MAIN_PRIO=0 python tools/exp_stream_priority.py
CRD_MAIN_PRIO=1 CRD_LATE_PRIO=-1 python tools/exp_stream_priority.py
