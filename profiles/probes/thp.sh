cd $GRAFT_REPO_ROOT && cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1 | head -2
bash profiles/probes/bam_e2e_only.sh
