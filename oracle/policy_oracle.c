/* placeholder translation unit, filled in with the bi-head policy restatement */
typedef int orc_policy_placeholder;
