"""CPU oracle for the W4/W3-A16 WQLinear hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``llm_awq_amd``) may import this package. Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.
"""
