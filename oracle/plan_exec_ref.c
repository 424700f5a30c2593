/* placeholder translation unit; the CPU reference executor for device plans is added with the
 * planner (test infrastructure only). */
int orc_plan_exec_abi(void) { return 0; }
