// placeholder, replaced later this round
extern "C" int oracle_mpc_placeholder(void) { return 0; }
