"""CPU oracle for the env.step hot path — TEST INFRASTRUCTURE ONLY (parity unpinned vs MuJoCo, see rg_oracle.c)."""
